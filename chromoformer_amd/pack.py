"""`python -m chromoformer_amd.pack` -- the binned genes of one metadata file as ONE memory-mappable file.

    python -m chromoformer_amd.pack -m data/E003/train.csv -d data/E003/npy -o data/E003/npy/chromoformer.cfstore

The reference re-reads ~9 `.npy` files per gene on every `__getitem__` (data.py:101-113; the files are written by
preprocessing/scripts/extract_signals.py:66-71, the metadata by prepare_train_metadata.py:87-121): ~170,000 `np.load`
calls per cell line before a process has seen its data once, repeated by every rank of a data-parallel run and by each
of the 44 jobs of the cross-validation sweep.  The packed store is built once per (cell line, binning configuration):
every gene of the metadata file is binned (on the GPU with cf_bin_regions when one is visible, else on the host) into the
compact layout the kernels consume -- 126 KB of fp32 features per gene plus centre-row pad masks -- and written behind a
JSON header as page-aligned arrays.  `train.py`, `predict.py` and the sweep pick it up (`--store`, or the default name
`<npy_dir>/chromoformer.cfstore`) when its signature matches the run: a split is then a row gather out of a memory map
(only the rank's shard under data parallelism), 18,000 genes in seconds.

File layout: 8 bytes magic "CFSTORE1", uint64 header length, JSON header, zero padding to 4,096, then the arrays named
in header["arrays"] = {name: [dtype, shape, byte offset behind the padded header]}: pf{r} [n,1,L,F] f32, cf{r} [n,S,L,F] f32, pm{r} [n,L] u8,
cm{r} [n,S,L] u8 (1 = padding), im [n,T,T] u8, freq [n,T,T] f32, label_cls [n] i64, label_reg [n] f32 (log2(expression
+ 1); NaN when the metadata has no expression column)."""
from __future__ import annotations

import argparse
import json
import os
import sys

import numpy as np
import pandas as pd
import torch

from .data import ChromoformerDataset, GeneStore

MAGIC = b"CFSTORE1"
DEFAULT_NAME = "chromoformer.cfstore"
_ALIGN = 4096


def signature(binsizes, i_max, w_prom, w_max, n_feats):
    return {"binsizes": [int(b) for b in binsizes], "i_max": int(i_max), "w_prom": int(w_prom), "w_max": int(w_max), "n_feats": int(n_feats)}


def gene_digests(table, npy_dir, genes=None):
    """{gene_id: digest} over everything of a metadata row that ends up in the store -- label, expression, TSS window, strand,
    partner regions and their scores -- and the byte sizes of the region files it names (a re-extracted signal file of another
    length changes it; contents are not re-read: that is what the store exists to avoid).  `table`: DataFrame of the metadata."""
    import hashlib
    want = None if genes is None else set(genes)
    out = {}
    cols = [c for c in ("label", "expression", "chrom", "start", "end", "strand", "neighbors", "scores") if c in table.columns]

    def size(name):
        try:
            return os.path.getsize(os.path.join(npy_dir, name + ".npy"))
        except OSError:
            return -1

    for r in table.to_dict("records"):
        g = r["gene_id"]
        if want is not None and g not in want:
            continue
        parts = []
        for c in cols:
            v = r[c]
            parts.append("" if isinstance(v, float) and np.isnan(v) else (repr(float(v)) if isinstance(v, (float, np.floating)) else str(v)))
        files = ["%s:%d-%d" % (r["chrom"], int(r["start"]) - 20000, int(r["start"]) + 20000)]
        if isinstance(r.get("neighbors"), str) and r["neighbors"]:
            files += r["neighbors"].split(";")
        parts += ["%s=%d" % (f, size(f)) for f in files]
        out[g] = hashlib.sha1("|".join(parts).encode()).hexdigest()[:20]
    return out


def pack(meta, npy_dir, out, binsizes=(2000, 500, 100), i_max=8, w_prom=40000, w_max=40000, n_feats=7, device="auto", progress=False):
    """Bin every gene of `meta` (file order) and write the packed store to `out`.  -> number of genes.
    device: "auto" = the current GPU when one is visible, None = bin on the host, or a torch.device."""
    table = pd.read_csv(meta)
    genes = table.gene_id.tolist()
    # metadata with only an `expression` column (a regression-only cell line) packs too: the class labels are then zeros
    ds = ChromoformerDataset(meta, npy_dir, genes, n_feats, i_max, list(binsizes), w_prom, w_max, regression="label" not in table.columns)
    if isinstance(device, str) and device == "auto":
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    store = GeneStore(ds, pin=False, progress=progress, device=device, resident=False) if device is not None else GeneStore(ds, pin=False, progress=progress)
    arrays = {}
    for r in range(len(store.binsizes)):
        arrays["pf%d" % r], arrays["cf%d" % r] = store.pf[r], store.cf[r]
        arrays["pm%d" % r], arrays["cm%d" % r] = store.pm[r], store.cm[r]
    arrays["im"], arrays["freq"], arrays["label_cls"] = store.im, store.freq, store.label
    if "expression" in table.columns:
        arrays["label_reg"] = torch.from_numpy(np.log2(table.expression.to_numpy(dtype=np.float64) + 1).astype(np.float32))
    else:
        arrays["label_reg"] = torch.full((len(genes),), float("nan"))
    if "label" not in table.columns:
        arrays["label_cls"] = torch.zeros(len(genes), dtype=torch.int64)
    write(out, genes, signature(binsizes, i_max, w_prom, w_max, n_feats), arrays, gene_digests(table, npy_dir))
    return len(genes)


def write(out, genes, sig, arrays, digests=None):
    """Write a packed store from already-binned arrays (torch tensors, any device).  `digests`: gene_digests() of the
    metadata the arrays were made from (a store without them is never picked up automatically)."""
    np_arrays = {k: v.detach().cpu().contiguous().numpy() for k, v in arrays.items()}
    header = {"version": 1, "n_genes": len(genes), "genes": list(genes), "signature": sig, "arrays": {}}
    if digests is not None:
        header["gene_sha"] = [digests[g] for g in genes]
    off = 0
    for k, a in np_arrays.items():                        # offsets are relative to the first page boundary behind the header
        header["arrays"][k] = [str(a.dtype), list(a.shape), off]
        off += (a.nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    blob = json.dumps(header).encode()
    base = (16 + len(blob) + _ALIGN - 1) // _ALIGN * _ALIGN
    tmp = out + ".tmp"
    with open(tmp, "wb") as f:
        f.write(MAGIC)
        f.write(np.uint64(len(blob)).tobytes())
        f.write(blob)
        f.write(b"\0" * (base - 16 - len(blob)))
        for k, a in np_arrays.items():
            assert f.tell() == base + header["arrays"][k][2]
            f.write(a.tobytes())
            f.write(b"\0" * (-a.nbytes % _ALIGN))
    os.replace(tmp, out)


class PackedStore:
    """A packed store opened read-only: header + one memory map per array (nothing is read until rows are gathered)."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            if f.read(8) != MAGIC:
                raise ValueError("%s is not a chromoformer packed store" % path)
            n = int(np.frombuffer(f.read(8), dtype=np.uint64)[0])
            self.header = json.loads(f.read(n).decode())
        self.genes = self.header["genes"]
        self.signature = self.header["signature"]
        self.gene_sha = dict(zip(self.genes, self.header["gene_sha"])) if "gene_sha" in self.header else None
        self._row = {g: i for i, g in enumerate(self.genes)}
        base = (16 + n + _ALIGN - 1) // _ALIGN * _ALIGN
        self.arrays = {k: (np.memmap(path, dtype=np.dtype(dt), mode="r", offset=base + off, shape=tuple(shape)) if int(np.prod(shape)) else
                           np.zeros(tuple(shape), np.dtype(dt)))
                       for k, (dt, shape, off) in self.header["arrays"].items()}

    def matches(self, binsizes, i_max, w_prom, w_max, n_feats):
        return self.signature == signature(binsizes, i_max, w_prom, w_max, n_feats)

    def stale(self, table, npy_dir, genes):
        """Genes (of `genes`) whose metadata row or signal-file sizes differ from what the store was packed from; all of
        them when the store carries no digests."""
        if self.gene_sha is None:
            return list(genes)
        now = gene_digests(table, npy_dir, genes)
        return [g for g in genes if self.gene_sha.get(g) != now.get(g)]

    def rows(self, genes):
        try:
            return np.asarray([self._row[g] for g in genes], dtype=np.int64)
        except KeyError as e:
            raise KeyError("gene %s is not in the packed store %s" % (e, self.path))

    def store(self, genes, device=None, regression=False):
        """GeneStore of `genes` (in that order): a row gather out of the memory maps, uploaded when `device` is given."""
        idx = self.rows(genes)
        order = np.argsort(idx, kind="stable")            # read the file front to back, then restore the requested order
        inv = np.empty_like(order)
        inv[order] = np.arange(len(order))

        inv_t = torch.from_numpy(inv)
        inv_d = inv_t.to(device) if device is not None else None

        def take(name):
            a = self.arrays[name]
            if not len(idx):
                t = torch.from_numpy(np.zeros((0,) + a.shape[1:], a.dtype))
                return t.to(device) if device is not None else t
            t = torch.from_numpy(np.ascontiguousarray(a[idx[order]]))      # one pass over the file, front to back
            if device is not None:                                          # upload, then restore the requested order on the device
                return (t.pin_memory() if torch.cuda.is_available() else t).to(device, non_blocking=True)[inv_d]
            return t[inv_t]

        nres = len(self.signature["binsizes"])
        lab = take("label_reg" if regression else "label_cls")
        if regression and len(idx) and bool(torch.isnan(lab).any()):
            raise ValueError("`expression` column is required for training ChromoformerRegression model.")
        st = GeneStore.from_arrays(self.signature["binsizes"], self.signature["i_max"], [take("pf%d" % r) for r in range(nres)],
                                   [take("cf%d" % r) for r in range(nres)], [take("pm%d" % r) for r in range(nres)],
                                   [take("cm%d" % r) for r in range(nres)], take("im"), take("freq"), lab, regression=regression)
        if device is not None:
            torch.cuda.synchronize(device)
        return st


last_skip_reason = None      # why the last find() returned None (train.py prints it: a skipped store means re-binning every .npy file)


def find(npy_dir, explicit, binsizes, i_max, w_prom, w_max, n_feats, genes=None, meta=None):
    """The packed store a run should use: `explicit` if given (must match, else an error), else `<npy_dir>/chromoformer.cfstore`
    when it exists, matches the binning configuration, holds all `genes` AND was packed from the metadata rows and signal
    files the run sees now (`meta`: the run's metadata file or DataFrame; labels, expression, partner sets, scores and TSS
    windows are frozen at pack time, so a store that no longer agrees with them is stale: an automatic pick-up skips it with
    a warning, an explicit --store is an error); None otherwise."""
    global last_skip_reason
    last_skip_reason = None
    path = explicit or os.path.join(npy_dir, DEFAULT_NAME)
    if not os.path.exists(path):
        if explicit:
            raise FileNotFoundError(explicit)
        last_skip_reason = "no %s" % path
        return None
    ps = PackedStore(path)
    ok = ps.matches(binsizes, i_max, w_prom, w_max, n_feats) and (genes is None or all(g in ps._row for g in genes))
    if not ok:
        if explicit:
            raise ValueError("%s was packed for %s, the run needs %s" % (path, ps.signature, signature(binsizes, i_max, w_prom, w_max, n_feats)))
        last_skip_reason = "%s was packed for another binning configuration or gene set" % path
        return None
    if meta is not None:
        table = pd.read_csv(meta) if isinstance(meta, (str, os.PathLike)) else meta
        stale = ps.stale(table, npy_dir, genes if genes is not None else table.gene_id.tolist())
        if stale:
            why = ("%s carries no content digests (packed by an older version)" % path if ps.gene_sha is None else
                   "%s is stale: the metadata rows / signal files of %d genes (first: %s) changed since it was packed" % (path, len(stale), stale[0]))
            if explicit:
                raise ValueError(why + "; re-run `python -m chromoformer_amd.pack`")
            import warnings
            warnings.warn(why + "; ignoring it and binning the raw .npy files (re-run `python -m chromoformer_amd.pack`)")
            last_skip_reason = why
            return None
    return ps


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("-m", "--meta", required=True)
    ap.add_argument("-d", "--npy-dir", required=True)
    ap.add_argument("-o", "--output", default=None, help="default: <npy-dir>/%s" % DEFAULT_NAME)
    ap.add_argument("--binsizes", nargs="+", type=int, default=[2000, 500, 100])
    ap.add_argument("--i-max", type=int, default=8)
    ap.add_argument("--w-prom", type=int, default=40000)
    ap.add_argument("--w-max", type=int, default=40000)
    ap.add_argument("--n-feats", type=int, default=7)
    ap.add_argument("--host", action="store_true", help="bin on the host even if a GPU is visible")
    args = ap.parse_args(argv)
    out = args.output or os.path.join(args.npy_dir, DEFAULT_NAME)
    dev = None if args.host else "auto"
    n = pack(args.meta, args.npy_dir, out, args.binsizes, args.i_max, args.w_prom, args.w_max, args.n_feats, device=dev, progress=True)
    print("packed %d genes -> %s (%.1f MB)" % (n, out, os.path.getsize(out) / 1e6))
    return 0


if __name__ == "__main__":
    sys.exit(main())
