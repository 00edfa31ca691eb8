"""Synthetic pre-binned batches for benchmarks ("Synthetic inputs"): reference batch layout
(`ChromoformerDataset.__getitem__` collated, data.py:205-212), features `log1p(Gamma(0.6, 1))` with 35 % zeros at
L >= 400, structured pad masks, sorted interaction frequencies in [1.5, 3].

    regime "dense"     every gene has i_max partners, every pCRE spans all bins (the worst case the benchmark is quoted on)
    regime "realistic" partner counts from the demo histogram, pCRE lengths ~ LogNormal(ln 5900, 0.5) clipped to [1800, w_max]

Product-side generator: bench.py and the tools use it, so that nothing but the parity checks and the CPU-baseline leg
touches oracle/.  tests/test_synth_cpu.py pins it to the oracle's generator (same seeds -> same batches)."""
from __future__ import annotations

import math

import numpy as np
import torch

_DEMO_PARTNER_HIST = (11, 8, 3, 3, 4, 5, 4, 4, 58)      # genes with 0..8 partners among the 100 demo genes


def synthetic_batch(B, seed=1234, regime="dense", regression=False, *, n_feats=7, i_max=8, binsizes=(2000, 500, 100), w_max=40000):
    rng = np.random.default_rng(seed)
    S, T = i_max, i_max + 1
    if regime == "dense":
        n_part = np.full(B, S)
    else:
        hist = np.asarray(_DEMO_PARTNER_HIST, dtype=np.float64) if S == 8 else np.ones(S + 1)
        n_part = rng.choice(S + 1, size=B, p=hist / hist.sum())
    lens = np.clip(rng.lognormal(np.log(5900), 0.5, size=(B, S)), 1800, w_max)
    has = np.arange(S)[None, :] < n_part[:, None]                       # [B, S] real partner?
    batch = {k: {} for k in ("promoter_feats", "promoter_pad_masks", "pcre_feats", "pcre_pad_masks", "interaction_masks")}
    for b in binsizes:
        L = w_max // b

        def draw(shape):
            x = np.log1p(rng.gamma(0.6, 1.0, size=shape)).astype(np.float32)
            if L >= 400:
                x[rng.random(shape) < 0.35] = 0.0
            return x

        pf, cf = draw((B, 1, L, n_feats)), draw((B, S, L, n_feats))
        n = np.full((B, S), L) if regime == "dense" else np.ceil(lens / b).astype(np.int64)
        lo = np.ceil((L - n) / 2).astype(np.int64)
        j = np.arange(L)[None, None, :]
        valid = has[:, :, None] & (j >= lo[:, :, None]) & (j < (lo + n)[:, :, None])      # [B, S, L] real pCRE bins
        cf *= valid[..., None]
        pm = np.zeros((B, 1, 1, L, L), dtype=bool)                       # promoters are never padded (w_prom = w_max)
        cm = np.broadcast_to(~valid[:, :, None, None, :], (B, S, 1, L, L)).copy()
        q = np.arange(T)
        im = ~((q[None, :, None] <= n_part[:, None, None]) & (q[None, None, :] <= n_part[:, None, None]))[:, None]
        for key, arr in (("promoter_feats", pf), ("promoter_pad_masks", pm), ("pcre_feats", cf), ("pcre_pad_masks", cm),
                         ("interaction_masks", np.ascontiguousarray(im))):
            batch[key][b] = torch.from_numpy(arr)
    freq = np.zeros((B, T, T), dtype=np.float32)
    for g in range(B):
        freq[g, 0, 1:1 + n_part[g]] = np.sort(rng.uniform(1.5, 3.0, size=n_part[g]))[::-1]
    batch["interaction_freq"] = torch.from_numpy(freq)
    if regression:
        batch["label"] = torch.from_numpy(rng.gamma(1.0, 2.0, size=B).astype(np.float32))
    else:
        batch["label"] = torch.from_numpy(rng.integers(0, 2, size=B).astype(np.int64))
    return batch


def synthetic_store(n_genes, device, seed=1234, regime="dense", regression=False, *, n_feats=7, i_max=8, binsizes=(2000, 500, 100),
                    w_max=40000, chunk=2048):
    """A resident GeneStore of `n_genes` synthetic genes drawn ON the device (same distributions as synthetic_batch, torch's
    generator instead of numpy's: 16,384 genes are 2 GB of features, too slow to draw on the host for a benchmark)."""
    from .data import GeneStore
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    S, T, G = i_max, i_max + 1, n_genes
    rnd = lambda *shape: torch.rand(*shape, device=device, generator=gen)
    if regime == "dense":
        n_part = torch.full((G,), S, device=device, dtype=torch.long)
    else:
        hist = torch.tensor(_DEMO_PARTNER_HIST if S == 8 else [1.0] * (S + 1), dtype=torch.float32, device=device)
        n_part = torch.multinomial(hist / hist.sum(), G, replacement=True, generator=gen)
    lens = torch.exp(math.log(5900.0) + 0.5 * torch.randn(G, S, device=device, generator=gen)).clamp(1800, w_max)
    has = torch.arange(S, device=device)[None, :] < n_part[:, None]
    pf, cf, pm, cm = [], [], [], []
    for b in binsizes:
        L = w_max // b

        def draw(shape):
            out = torch.empty(shape, device=device)
            for g0 in range(0, shape[0], chunk):
                sl = out[g0:g0 + chunk]
                x = torch.log1p(torch._standard_gamma(torch.full(sl.shape, 0.6, device=device), generator=gen))
                if L >= 400:
                    x[rnd(*sl.shape) < 0.35] = 0.0
                sl.copy_(x)
            return out

        n = torch.full((G, S), L, device=device, dtype=torch.long) if regime == "dense" else torch.ceil(lens / b).long()
        lo = torch.ceil((L - n).float() / 2).long()
        j = torch.arange(L, device=device)[None, None, :]
        valid = has[:, :, None] & (j >= lo[:, :, None]) & (j < (lo + n)[:, :, None])
        c = draw((G, S, L, n_feats))
        c *= valid[..., None]
        pf.append(draw((G, 1, L, n_feats)))
        cf.append(c)
        pm.append(torch.zeros(G, L, dtype=torch.uint8, device=device))
        cm.append((~valid).to(torch.uint8).contiguous())
    q = torch.arange(T, device=device)
    im = (~((q[None, :, None] <= n_part[:, None, None]) & (q[None, None, :] <= n_part[:, None, None]))).to(torch.uint8).contiguous()
    freq = torch.zeros(G, T, T, device=device)
    sc = torch.sort(1.5 + 1.5 * rnd(G, S), dim=1, descending=True).values * has
    freq[:, 0, 1:] = sc
    label = (4.0 * rnd(G)) if regression else (rnd(G) < 0.5).long()
    return GeneStore.from_arrays(binsizes, i_max, pf, cf, pm, cm, im, freq, label, regression=regression)
