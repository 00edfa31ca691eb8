"""CPU oracle for the per-gene input pipeline  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates what the reference's ``ChromoformerDataset.__getitem__`` produces
(/root/reference/chromoformer/data.py:68-212) as vectorised numpy/torch, written
from its behaviour.  Pinned by golden G5 (tests/golden/make_goldens.py): full
``__getitem__`` outputs of the imported reference on synthetic raw regions,
including a '-' strand gene, a gene with no partners and a partial last bin.

  bin_and_pad ............ data.py:68-99   (mean per bin incl. the short tail bin,
                                           natural log1p, ceil-left / floor-right pad)
  region ................. data.py:101-113 (centre window slice, strand flip)
  gene_item .............. data.py:115-212
"""
from __future__ import annotations

import math

import numpy as np
import torch


def bin_and_pad(x, bin_size, max_n_bins):
    """x: float32 [F, len] -> ([F, max_n_bins], left_pad, n_bins, right_pad)."""
    x = torch.as_tensor(x).float()
    F_, length = x.shape
    n_full, tail = divmod(length, bin_size)
    cols = []
    if n_full:
        cols.append(x[:, :n_full * bin_size].reshape(F_, n_full, bin_size).mean(dim=2))
    if tail:
        cols.append(x[:, n_full * bin_size:].mean(dim=1, keepdim=True))
    binned = torch.log(torch.cat(cols, dim=1) + 1)
    n_bins = binned.shape[1]
    left = math.ceil((max_n_bins - n_bins) / 2)
    right = (max_n_bins - n_bins) // 2
    out = torch.zeros(F_, max_n_bins)
    out[:, left:left + n_bins] = binned
    return out, left, n_bins, right


def region(raw, bin_size, max_n_bins, strand="+", window=None):
    x = torch.as_tensor(np.asarray(raw)).float()
    if window is not None:
        x = x[:, 20000 - window // 2: 20000 + window // 2]
    out, left, n_bins, right = bin_and_pad(x, bin_size, max_n_bins)
    if strand == "+":
        return out, left, n_bins, right
    return torch.flip(out, dims=[1]), right, n_bins, left


def gene_item(load, tss, strand, pcres, scores, label, *, i_max=8, binsizes=(2000, 500, 100),
              w_prom=40000, w_max=40000, n_feats=7, regression=False):
    """``load(chrom, start, end)`` returns the raw [F, len] array of a region.

    tss = (chrom, start); pcres = [(chrom, start, end), ...]."""
    chrom, start = tss
    T = i_max + 1
    item = {"label": torch.tensor(label).float() if regression else torch.tensor(label).long()}
    for k in ("promoter_feats", "promoter_pad_masks", "pcre_feats", "pcre_pad_masks", "interaction_masks"):
        item[k] = {}
    raw_p = load(chrom, start - 20000, start + 20000)
    raws_c = [load(*p) for p in pcres]
    n_part = len(pcres)
    for b in binsizes:
        L = w_max // b
        xp, lo_p, n_p, _ = region(raw_p, b, L, strand, window=w_prom)
        mp = torch.ones(1, 1, L, L, dtype=torch.bool)
        mp[0, 0, lo_p:lo_p + n_p, lo_p:lo_p + n_p] = False
        cf = torch.zeros(i_max, L, n_feats)
        cm = torch.ones(i_max, 1, L, L, dtype=torch.bool)
        for i, raw in enumerate(raws_c):
            xc, lo_c, n_c, _ = region(raw, b, L)
            cf[i] = xc.t()
            cm[i, 0, lo_p:lo_p + n_p, lo_c:lo_c + n_c] = False
        im = torch.ones(1, T, T, dtype=torch.bool)
        im[0, :n_part + 1, :n_part + 1] = False
        item["promoter_feats"][b] = xp.t().unsqueeze(0)
        item["promoter_pad_masks"][b] = mp
        item["pcre_feats"][b] = cf
        item["pcre_pad_masks"][b] = cm
        item["interaction_masks"][b] = im
    freq = torch.zeros(T, T)
    for i, s in enumerate(scores):
        freq[0, i + 1] = s
    item["interaction_freq"] = freq
    return item
