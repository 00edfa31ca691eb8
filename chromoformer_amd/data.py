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


#: layout of cf_bin_job (include/chromoformer_hip.h)
BIN_JOB = np.dtype([("raw", "<u8"), ("ld", "<i8"), ("col0", "<i4"), ("ncols", "<i4"), ("flip", "<i4"), ("reserved", "<i4"),
                    ("out", "<u8"), ("mask", "<u8")])


#: layout of cf_bin_job_multi: out / mask per resolution, coarsest resolution first
BIN_JOB_MULTI = np.dtype([("raw", "<u8"), ("ld", "<i8"), ("col0", "<i4"), ("ncols", "<i4"), ("flip", "<i4"), ("reserved", "<i4"),
                          ("out", "<u8", (3,)), ("mask", "<u8", (3,))])


class GeneStore:
    """All genes of a split, binned once, compact layout.

    ``device=None``: binned on the host (numpy/torch), kept in pinned host memory.
    ``device=<cuda device>``: the raw fp16 signals are shipped to the GPU and binned there by ``cf_bin_regions``
    (the HIP replacement of data.py:68-113), straight into the store arrays; with ``resident=True`` the store then
    stays in HBM (126 KB per gene: 18,000 genes = 2.3 GB of the 288 GB) and batches are gathered on the device,
    otherwise it is copied back into pinned host memory and batches travel per step as before."""

    def __init__(self, dataset, pin=None, progress=False, device=None, resident=False, chunk_bytes=1 << 29):
        ds = dataset
        G, S, T, F = len(ds), ds.i_max, ds.i_max + 1, ds.n_feats
        self.binsizes, self.i_max, self.regression = ds.binsizes, S, ds.regression
        self.n_bins = [ds.w_max // b for b in ds.binsizes]
        self.n = G
        pin = torch.cuda.is_available() if pin is None else pin

        def arena(shape, dtype, fill=0, dev=None):
            if dev is not None:
                return torch.full(shape, fill, dtype=dtype, device=dev)
            t = torch.full(shape, fill, dtype=dtype)
            return t.pin_memory() if pin else t

        self.im = arena((G, T, T), torch.uint8, 1)
        self.freq = arena((G, T, T), torch.float32)
        self.label = arena((G,), torch.float32 if ds.regression else torch.int64)
        for i in range(G):
            g = ds.genes[ds.target_genes[i]]
            n_part = len(g["pcres"])
            self.im[i, :n_part + 1, :n_part + 1] = 0
            for s, sc in enumerate(g["scores"]):
                self.freq[i, 0, s + 1] = sc
            self.label[i] = float(g["label"]) if ds.regression else int(g["label"])
        if device is not None:
            self._bin_on_device(ds, torch.device(device), arena, resident, chunk_bytes, progress)
            return
        self.pf = [arena((G, 1, L, F), torch.float32) for L in self.n_bins]
        self.cf = [arena((G, S, L, F), torch.float32) for L in self.n_bins]
        self.pm = [arena((G, L), torch.uint8) for L in self.n_bins]
        self.cm = [arena((G, S, L), torch.uint8) for L in self.n_bins]
        it = range(G)
        if progress:
            from tqdm import tqdm
            it = tqdm(it, desc="binning")
        for i in it:
            gene = ds.target_genes[i]
            for r, (b, (p, lo_p, n_p, pcs)) in enumerate(ds.regions(gene).items()):
                self.pf[r][i, 0] = p.t()
                self.pm[r][i] = 1
                self.pm[r][i, lo_p:lo_p + n_p] = 0
                self.cm[r][i] = 1                       # dummy slots: fully masked (data.py:196-198)
                for s, (x, lo, n) in enumerate(pcs):
                    self.cf[r][i, s] = x.t()
                    self.cm[r][i, s, lo:lo + n] = 0

    def _bin_on_device(self, ds, dev, arena, resident, chunk_bytes, progress):
        """Raw fp16 regions -> HBM -> cf_bin_regions -> store arrays (on the device)."""
        import ctypes as C
        from . import _lib
        L_ = _lib.lib()
        G, S, F = len(ds), ds.i_max, ds.n_feats
        pf = [arena((G, 1, L, F), torch.float32, 0, dev) for L in self.n_bins]
        cf = [arena((G, S, L, F), torch.float32, 0, dev) for L in self.n_bins]
        pm = [arena((G, L), torch.uint8, 1, dev) for L in self.n_bins]
        cm = [arena((G, S, L), torch.uint8, 1, dev) for L in self.n_bins]       # dummy slots stay fully masked
        col0 = 20000 - ds.w_prom // 2                                           # data.py:106-107
        if col0 < 0:
            raise ValueError("w_prom = %d exceeds the 40,000-sample promoter files" % ds.w_prom)
        stream = torch.cuda.current_stream(dev)
        it = range(G)
        if progress:
            from tqdm import tqdm
            it = tqdm(it, desc="loading raw signals")
        pending, pending_bytes = [], 0

        def flush():
            nonlocal pending, pending_bytes
            if not pending:
                return
            flat = torch.from_numpy(np.concatenate([a.reshape(-1) for _, _, _, a in pending]))
            raw = (flat.pin_memory() if torch.cuda.is_available() else flat).to(dev, non_blocking=True)
            base, off = raw.data_ptr(), 0
            # every resolution of a region in ONE launch (cf_bin_regions_multi: nested bin sizes read the raw bytes once); more than
            # three resolutions or repeated bin sizes: one cf_bin_regions launch per resolution
            order = sorted(range(len(self.binsizes)), key=lambda r: -self.binsizes[r])      # coarsest first
            multi = len(self.binsizes) <= 3 and len(set(self.binsizes)) == len(self.binsizes)
            jobs = [[] for _ in self.binsizes]
            mjobs = np.zeros(len(pending), dtype=BIN_JOB_MULTI)
            max_cols = 0
            for k, (i, s, flip, a) in enumerate(pending):
                ld = a.shape[1]
                c0, nc = (col0, max(0, min(col0 + ds.w_prom, ld) - col0)) if s < 0 else (0, ld)
                max_cols = max(max_cols, nc)
                mjobs[k]["raw"], mjobs[k]["ld"], mjobs[k]["col0"], mjobs[k]["ncols"], mjobs[k]["flip"] = base + 2 * off, ld, c0, nc, int(flip)
                for r, b in enumerate(self.binsizes):
                    L = self.n_bins[r]
                    if -(-nc // b) > L:
                        raise ValueError("region spans %d bins but w_max allows %d" % (-(-nc // b), L))
                    out = pf[r][i, 0] if s < 0 else cf[r][i, s]
                    msk = pm[r][i] if s < 0 else cm[r][i, s]
                    if multi:
                        mjobs[k]["out"][order.index(r)], mjobs[k]["mask"][order.index(r)] = out.data_ptr(), msk.data_ptr()
                    else:
                        jobs[r].append((base + 2 * off, ld, c0, nc, int(flip), 0, out.data_ptr(), msk.data_ptr()))
                off += a.size
            if multi:
                tab = torch.from_numpy(mjobs.view(np.uint8)).to(dev)
                nr = len(order)
                bs = (C.c_int * nr)(*[self.binsizes[r] for r in order])
                nb = (C.c_int * nr)(*[self.n_bins[r] for r in order])
                _lib.check(L_.cf_bin_regions_multi(C.c_void_p(tab.data_ptr()), len(pending), F, nr, bs, nb, int(max_cols), stream.cuda_stream),
                           "cf_bin_regions_multi")
            else:
                for r, b in enumerate(self.binsizes):
                    tab = torch.from_numpy(np.array(jobs[r], dtype=BIN_JOB).view(np.uint8)).to(dev)
                    _lib.check(L_.cf_bin_regions(C.c_void_p(tab.data_ptr()), len(jobs[r]), F, b, self.n_bins[r], stream.cuda_stream),
                               "cf_bin_regions")
            stream.synchronize()                                                # raw / job tables may be released
            pending, pending_bytes = [], 0

        for i in it:
            g = ds.genes[ds.target_genes[i]]
            chrom, tss, strand = g["tss"]
            regions = [(-1, strand != "+", ds._load(chrom, tss - 20000, tss + 20000))] + [(s, False, ds._load(*p)) for s, p in enumerate(g["pcres"])]
            for s, flip, a in regions:
                if a.dtype != np.float16:
                    # the reference widens whatever dtype the file has (data.py:104); the device path ships fp16, which is
                    # what preprocessing writes (extract_signals.py:66-71) -- refuse a lossy narrowing instead of training on inf
                    h = a.astype(np.float16)
                    if not np.array_equal(h.astype(a.dtype), a):
                        raise ValueError("raw signal file is %s and does not fit float16 exactly: bin it on the host "
                                         "(GeneStore(device=None)) or re-save it as float16" % a.dtype)
                    a = h
                a = np.ascontiguousarray(a, dtype=np.float16)
                if a.shape[0] != F:
                    raise ValueError("expected %d feature rows, file has %d" % (F, a.shape[0]))
                pending.append((i, s, flip, a))
                pending_bytes += a.nbytes
            if pending_bytes >= chunk_bytes:
                flush()
        flush()
        if resident:
            self.pf, self.cf, self.pm, self.cm = pf, cf, pm, cm
            self.im, self.freq, self.label = self.im.to(dev), self.freq.to(dev), self.label.to(dev)
        else:
            host = lambda t: (t.cpu().pin_memory() if torch.cuda.is_available() else t.cpu())
            self.pf, self.cf, self.pm, self.cm = ([host(t) for t in x] for x in (pf, cf, pm, cm))
        self.device = dev if resident else None

    @classmethod
    def from_arrays(cls, binsizes, i_max, pf, cf, pm, cm, im, freq, label, regression=False):
        """A store around arrays that are already binned and resident (compact layout of `_bin_on_device`): per
        resolution pf [n,1,L,F], cf [n,S,L,F], pm [n,L], cm [n,S,L] (uint8, 1 = padding); im [n,T,T] uint8, freq
        [n,T,T], label [n].  Used by the benchmark's synthetic split and by the packed-store loader."""
        self = cls.__new__(cls)
        self.binsizes, self.i_max, self.regression = list(binsizes), i_max, regression
        self.n_bins = [t.shape[2] for t in pf]
        self.n = int(freq.shape[0])
        self.pf, self.cf, self.pm, self.cm, self.im, self.freq, self.label = pf, cf, pm, cm, im, freq, label
        self.device = freq.device if freq.is_cuda else None
        return self

    def __len__(self):
        return self.n

    def struct(self):
        """cf_store view of a device-resident store (include/chromoformer_hip.h), for cf_gather_batch."""
        from . import _lib
        if getattr(self, "device", None) is None:
            raise RuntimeError("GeneStore.struct(): the store is not resident on a device")
        st = _lib.cf_store()
        st.n_genes = self.n
        for r in range(len(self.binsizes)):
            for t in (self.pf[r], self.cf[r], self.pm[r], self.cm[r]):
                if not t.is_contiguous():
                    raise RuntimeError("GeneStore arrays must be contiguous")
            st.promoter_feats[r], st.pcre_feats[r] = self.pf[r].data_ptr(), self.cf[r].data_ptr()
            st.promoter_mask[r], st.pcre_mask[r] = self.pm[r].data_ptr(), self.cm[r].data_ptr()
        st.interaction_mask, st.interaction_freq, st.labels = self.im.data_ptr(), self.freq.data_ptr(), self.label.data_ptr()
        return st

    def batch(self, idx):
        """Gather genes `idx` into a batch dict the device Slot understands (compact masks)."""
        idx = torch.as_tensor(idx, dtype=torch.long)
        if getattr(self, "device", None) is not None:
            idx = idx.to(self.device)                   # resident store: the gather runs on the device
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


def static_shard(genes, rank, world):
    """The genes a data-parallel rank owns (and loads): every world-th gene of the split, so that shards differ by at most
    one gene and each is as mixed as the whole."""
    return list(genes[rank::world])


def static_epoch_batches(perm, rank, world, batch_size):
    """Batches of a rank under static sharding, as row indices into ITS shard (static_shard): the epoch permutation of the
    whole split -- the same on every rank -- is filtered to the rank's own genes (gene p is row p // world of shard
    p % world), and the first floor(floor(n / world) / batch_size) batches are kept, a count every rank agrees on.
    A global batch is then the union of one batch per rank: `batch_size` genes from each shard."""
    own = [p // world for p in perm if p % world == rank]
    n_b = (len(perm) // world) // batch_size
    return [own[k * batch_size:(k + 1) * batch_size] for k in range(n_b)]
