"""Full-scale epoch through the shipped entrypoint: a synthetic Roadmap-sized cell line (18,955 genes -- the gene count of a
Roadmap train.csv --, partner counts and pCRE lengths drawn from the demo histograms: synth.synthetic_store(regime="realistic"))
packed once, then `python -m chromoformer_amd.train --store ... --timing` for three epochs, each with validation + checkpoint, with the
wall-clock breakdown train.py prints and the step rate of the epoch against bench.py's `train_loop` figure.
    python tools/epoch_evidence.py [--genes 18955] [--bench-genes-per-s 116000]
The raw signal files of such a cell line would be ~17 GB; the store is written from already-binned synthetic arrays (pack.write),
which is what `python -m chromoformer_amd.pack` produces from real files (tests/test_pack_gpu.py covers that path)."""
import argparse, os, re, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, pandas as pd, torch, yaml

ap = argparse.ArgumentParser()
ap.add_argument("--genes", type=int, default=18955)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--bench-genes-per-s", type=float, default=0.0, help="bench.py train_loop value to compare the epoch's step rate with")
a = ap.parse_args()
from chromoformer_amd import pack
from chromoformer_amd.synth import synthetic_store
n = a.genes
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
t0 = time.perf_counter()
st = synthetic_store(n, dev, seed=1, regime="realistic")
genes = ["ENSGSYN%05d" % i for i in range(n)]
rng = np.random.default_rng(0)
label = st.label.cpu().numpy().astype(int)
meta = pd.DataFrame(dict(gene_id=genes, expression=np.round(rng.gamma(2.0, 2.0, n) * np.where(label > 0, 2.5, 0.2), 3), eid="E000", label=label,
                         chrom=["chr%d" % (1 + i % 22) for i in range(n)], start=1_000_000 + 150_000 * np.arange(n), end=1_000_001 + 150_000 * np.arange(n),
                         strand=np.where(np.arange(n) % 2 == 0, "+", "-"), split=1 + np.arange(n) % 4, neighbors=np.nan, scores=np.nan))
os.makedirs(os.path.join(d, "npy"))
meta_path = os.path.join(d, "npy", "train.csv")
meta.to_csv(meta_path, index=False)
arrays = {}
for r in range(3):
    arrays["pf%d" % r], arrays["cf%d" % r], arrays["pm%d" % r], arrays["cm%d" % r] = st.pf[r], st.cf[r], st.pm[r], st.cm[r]
arrays["im"], arrays["freq"], arrays["label_cls"] = st.im, st.freq, st.label
arrays["label_reg"] = torch.from_numpy(np.log2(meta.expression.to_numpy(dtype=np.float64) + 1).astype(np.float32))
store = os.path.join(d, "npy", pack.DEFAULT_NAME)
pack.write(store, genes, pack.signature([2000, 500, 100], 8, 40000, 40000, 7), arrays, pack.gene_digests(pd.read_csv(meta_path), os.path.join(d, "npy")))
print("synthetic cell line: %d genes, packed store %.2f GB written in %.1f s" % (n, os.path.getsize(store) / 1e9, time.perf_counter() - t0))
del st, arrays
torch.cuda.empty_cache()
cfg = yaml.safe_load(open(os.path.join(ROOT, "chromoformer_amd", "configs", "default.yaml")))
cfg["bsz"], cfg["num_epoch"] = 64, 1 + a.epochs
yaml.safe_dump(cfg, open(os.path.join(d, "cfg.yaml"), "w"))
t0 = time.perf_counter()
r = subprocess.run([sys.executable, "-m", "chromoformer_amd.train", "-o", os.path.join(d, "ck.pt"), "-c", os.path.join(d, "cfg.yaml"), "--exp-id", "epoch",
                    "-m", meta_path, "-d", os.path.join(d, "npy"), "--fold", "0", "--timing"], cwd=ROOT, capture_output=True, text=True)
wall = time.perf_counter() - t0
out = r.stdout + r.stderr
print("python -m chromoformer_amd.train: rc %d, %.1f s wall including interpreter start and imports" % (r.returncode, wall))
keep = False
for ln in out.splitlines():
    if ln.startswith("packed store:") or ln.startswith("Validation") or "wall-clock breakdown" in ln:
        keep = keep or "wall-clock" in ln
        print("  " + ln.strip()[:200])
    elif keep and ln.startswith("  "):
        print("  " + ln.rstrip()[:200])
rows = re.findall(r"([0-9.]+) s\s+[0-9.]+ %\s+epoch (\d+): ([a-z0-9 -]+?)(?: \(|$)", out, re.M)
steps_n = {int(e): int(n) for e, n in re.findall(r"epoch (\d+): (\d+) steps", out)}
per = {}
for sec, e, what in rows:
    per.setdefault(int(e), {})[what.strip().split(" (")[0]] = float(sec)
for e in sorted(per):
    d_ = per[e]
    st_ = next((v for k, v in d_.items() if k.endswith("steps")), None)
    if st_ is None or e not in steps_n:
        continue
    gpu = st_ + d_.get("validation forward", 0.0)
    host = d_.get("validation metrics", 0.0) + d_.get("checkpoint hand-off", 0.0)
    rate = 64 * steps_n[e] / st_
    line = "epoch %d: %d steps in %.3f s = %.1f genes/s (%.4f ms per step; permutation, begin_epoch, running metrics%s included)" % (
        e, steps_n[e], st_, rate, 1e3 * st_ / steps_n[e], ", first-step validation + graph capture" if e == 1 else "")
    if a.bench_genes_per_s > 0:
        line += "; vs train_loop %.1f: %+.1f %%" % (a.bench_genes_per_s, 100 * (rate / a.bench_genes_per_s - 1))
    print(line)
    print("         host-side share (validation metrics + checkpoint hand-off): %.3f s of the epoch's %.3f s = %.1f %%" % (host, gpu + host, 100 * host / (gpu + host)))
if r.returncode:
    print(out[-3000:])
