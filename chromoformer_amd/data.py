"""Per-gene input pipeline.

``ChromoformerDataset`` mirrors the reference class (/root/reference/chromoformer/data.py:
23-215): same constructor, same ``__getitem__`` dictionary (features, bool pad masks,
interaction mask / frequencies, label).  It exists for drop-in compatibility and for the
parity tests; the training path does not go through it per step.

``GeneStore`` is what replaces the reference's DataLoader (train.py:137-140): every gene is
binned ONCE into a pinned host arena in the compact layout the HIP kernels consume
(126 KB of features per gene + the centre-row pad masks as bytes instead of 1.5 MB of
L x L bool masks), and batches are gathered from it and shipped to a device ``Slot`` with one
asynchronous copy per array (hipMemcpyAsync from pinned memory).
"""
from __future__ import annotations

import math
import os

import numpy as np
import pandas as pd
import torch


def _split_interval(s):
    chrom, rest = s.split(":")
    a, b = rest.split("-")
    return chrom, int(a), int(b)


def bin_log1p(x, bin_size):
    """[F, len] float32 -> [F, ceil(len/bin)]: per-bin mean (the last bin may be short, data.py:80-81)
    followed by natural log(1 + x) (data.py:82)."""
    x = torch.as_tensor(x, dtype=torch.float32)
    n_full, tail = divmod(x.shape[1], bin_size)
    parts = []
    if n_full:
        parts.append(x[:, :n_full * bin_size].reshape(x.shape[0], n_full, bin_size).mean(dim=2))
    if tail:
        parts.append(x[:, n_full * bin_size:].mean(dim=1, keepdim=True))
    return torch.log(torch.cat(parts, dim=1) + 1)


def centred(binned, L, flip=False):
    """Zero-pad to L bins with ceil(left) / floor(right) padding (data.py:87-88); a '-' strand
    region is mirrored after padding, which swaps the pads (data.py:110-113).
    -> ([F, L], first valid bin, number of valid bins)"""
    n = binned.shape[1]
    if n > L:
        raise ValueError("region spans %d bins but w_max allows %d" % (n, L))
    left = math.ceil((L - n) / 2)
    out = torch.zeros(binned.shape[0], L)
    out[:, left:left + n] = binned
    if flip:
        return torch.flip(out, dims=[1]), (L - n) // 2, n
    return out, left, n


class ChromoformerDataset(torch.utils.data.Dataset):
    def __init__(self, meta, npy_dir, target_genes, n_feats=7, i_max=8, binsizes=[2000, 500, 100],
                 w_prom=40000, w_max=40000, regression=False):
        super().__init__()
        self.npy_dir, self.n_feats = npy_dir, n_feats
        self.target_genes = list(target_genes)
        self.meta = pd.read_csv(meta) if isinstance(meta, (str, os.PathLike)) else meta
        self.regression = regression
        self.i_max, self.w_prom, self.w_max = i_max, w_prom, w_max
        self.binsizes = [int(b) for b in binsizes]          # the reference crashes on CLI strings (train.py:33)
        self.genes = {}
        for r in self.meta.to_dict("records"):
            has = isinstance(r.get("neighbors"), str) and r["neighbors"] != ""
            self.genes[r["gene_id"]] = dict(
                tss=(r["chrom"], int(r["start"]), r["strand"]),
                pcres=[_split_interval(s) for s in r["neighbors"].split(";")] if has else [],
                scores=[float(s) for s in str(r["scores"]).split(";")] if has else [],
                label=(np.log2(r["expression"] + 1) if regression else r["label"]),
            )

    def __len__(self):
        return len(self.target_genes)

    # ------------------------------------------------------------------ raw regions
    def _load(self, chrom, start, end):
        return np.load("%s/%s:%d-%d.npy" % (self.npy_dir, chrom, start, end))

    def regions(self, gene):
        """Binned + centred regions of a gene for every resolution:
        {binsize: (promoter [F,L], lo_p, n_p, [(pcre [F,L], lo, n), ...])}"""
        g = self.genes[gene]
        chrom, tss, strand = g["tss"]
        raw_p = torch.as_tensor(self._load(chrom, tss - 20000, tss + 20000)).float()
        raw_p = raw_p[:, 20000 - self.w_prom // 2: 20000 + self.w_prom // 2]     # data.py:106-107
        raws = [torch.as_tensor(self._load(*p)).float() for p in g["pcres"]]
        out = {}
        for b in self.binsizes:
            L = self.w_max // b
            p, lo_p, n_p = centred(bin_log1p(raw_p, b), L, flip=(strand != "+"))
            out[b] = (p, lo_p, n_p, [centred(bin_log1p(x, b), L) for x in raws])
        return out

    # ------------------------------------------------------------------ reference-format item
    def __getitem__(self, i):
        gene = self.target_genes[i]
        g = self.genes[gene]
        S, T = self.i_max, self.i_max + 1
        n_part = len(g["pcres"])
        item = {"label": torch.tensor(g["label"]).float() if self.regression else torch.tensor(g["label"]).long()}
        for k in ("promoter_feats", "promoter_pad_masks", "pcre_feats", "pcre_pad_masks", "interaction_masks"):
            item[k] = {}
        for b, (p, lo_p, n_p, pcs) in self.regions(gene).items():
            L = p.shape[1]
            rows = torch.ones(L, dtype=torch.bool)
            rows[lo_p:lo_p + n_p] = False
            mp = rows[:, None] | rows[None, :]
            feats = torch.zeros(S, L, self.n_feats)
            masks = torch.ones(S, 1, L, L, dtype=torch.bool)
            for s, (x, lo, n) in enumerate(pcs):
                feats[s] = x.t()
                cols = torch.ones(L, dtype=torch.bool)
                cols[lo:lo + n] = False
                masks[s, 0] = rows[:, None] | cols[None, :]
            im = torch.ones(1, T, T, dtype=torch.bool)
            im[0, :n_part + 1, :n_part + 1] = False
            item["promoter_feats"][b] = p.t().unsqueeze(0)
            item["promoter_pad_masks"][b] = mp[None, None]
            item["pcre_feats"][b] = feats
            item["pcre_pad_masks"][b] = masks
            item["interaction_masks"][b] = im
        freq = torch.zeros(T, T)
        for s, sc in enumerate(g["scores"]):
            freq[0, s + 1] = sc
        item["interaction_freq"] = freq
        return item


class GeneStore:
    """All genes of a split, binned once, in pinned host memory, compact layout."""

    def __init__(self, dataset, pin=None, progress=False):
        ds = dataset
        G, S, T, F = len(ds), ds.i_max, ds.i_max + 1, ds.n_feats
        self.binsizes, self.i_max, self.regression = ds.binsizes, S, ds.regression
        self.n_bins = [ds.w_max // b for b in ds.binsizes]
        pin = torch.cuda.is_available() if pin is None else pin

        def arena(shape, dtype):
            t = torch.zeros(shape, dtype=dtype)
            return t.pin_memory() if pin else t

        self.pf = [arena((G, 1, L, F), torch.float32) for L in self.n_bins]
        self.cf = [arena((G, S, L, F), torch.float32) for L in self.n_bins]
        self.pm = [arena((G, L), torch.uint8) for L in self.n_bins]
        self.cm = [arena((G, S, L), torch.uint8) for L in self.n_bins]
        self.im = arena((G, T, T), torch.uint8)
        self.freq = arena((G, T, T), torch.float32)
        self.label = arena((G,), torch.float32 if ds.regression else torch.int64)
        it = range(G)
        if progress:
            from tqdm import tqdm
            it = tqdm(it, desc="binning")
        for i in it:
            gene = ds.target_genes[i]
            g = ds.genes[gene]
            n_part = len(g["pcres"])
            for r, (b, (p, lo_p, n_p, pcs)) in enumerate(ds.regions(gene).items()):
                self.pf[r][i, 0] = p.t()
                self.pm[r][i] = 1
                self.pm[r][i, lo_p:lo_p + n_p] = 0
                self.cm[r][i] = 1                       # dummy slots: fully masked (data.py:196-198)
                for s, (x, lo, n) in enumerate(pcs):
                    self.cf[r][i, s] = x.t()
                    self.cm[r][i, s, lo:lo + n] = 0
            self.im[i] = 1
            self.im[i, :n_part + 1, :n_part + 1] = 0
            for s, sc in enumerate(g["scores"]):
                self.freq[i, 0, s + 1] = sc
            self.label[i] = float(g["label"]) if ds.regression else int(g["label"])
        self.n = G

    def __len__(self):
        return self.n

    def batch(self, idx):
        """Gather genes `idx` into a batch dict the device Slot understands (compact masks)."""
        idx = torch.as_tensor(idx, dtype=torch.long)
        d = {k: {} for k in ("promoter_feats", "promoter_pad_masks", "pcre_feats", "pcre_pad_masks", "interaction_masks")}
        for r, b in enumerate(self.binsizes):
            d["promoter_feats"][b] = self.pf[r][idx]
            d["pcre_feats"][b] = self.cf[r][idx]
            d["promoter_pad_masks"][b] = self.pm[r][idx]
            d["pcre_pad_masks"][b] = self.cm[r][idx]
            d["interaction_masks"][b] = self.im[idx]
        d["interaction_freq"] = self.freq[idx]
        d["label"] = self.label[idx]
        return d


def shard_indices(perm, rank, world, batch_size, drop_last=True):
    """Rank-sharded batches of an epoch permutation: global batch k is perm[k*G:(k+1)*G] with
    G = world * batch_size, rank r takes the r-th slice of it (so that averaging the per-rank
    mean-loss gradients reproduces the single-process gradient of the global batch).  With
    drop_last the tail that does not fill a global batch is dropped (train.py:138)."""
    G = world * batch_size
    n = len(perm)
    out = []
    full = n // G
    for k in range(full):
        out.append(perm[k * G + rank * batch_size: k * G + (rank + 1) * batch_size])
    if not drop_last and n % G:
        tail = perm[full * G:]
        per = math.ceil(len(tail) / world)
        out.append(tail[rank * per:(rank + 1) * per])
    return out
