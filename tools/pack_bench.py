"""Start-up of a training process from a packed store: 18,000 synthetic genes written once, then opened and gathered into
a resident split the way train.py does ("packed per-cell-line store built once").
    python tools/pack_bench.py [n_genes]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chromoformer_amd import pack
from chromoformer_amd.synth import synthetic_store
n = int(sys.argv[1]) if len(sys.argv) > 1 else 18000
dev = torch.device("cuda", 0)
st = synthetic_store(n, dev, seed=1, regime="realistic")
arrays = {}
for r in range(3):
    arrays["pf%d" % r], arrays["cf%d" % r], arrays["pm%d" % r], arrays["cm%d" % r] = st.pf[r], st.cf[r], st.pm[r], st.cm[r]
arrays["im"], arrays["freq"], arrays["label_cls"], arrays["label_reg"] = st.im, st.freq, st.label, torch.zeros(n)
genes = ["G%05d" % i for i in range(n)]
path = os.path.join(tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp")), "synth.cfstore")
t0 = time.perf_counter()
pack.write(path, genes, pack.signature([2000, 500, 100], 8, 40000, 40000, 7), arrays)
print("written %.2f GB in %.1f s" % (os.path.getsize(path) / 1e9, time.perf_counter() - t0))
del st, arrays
rng = np.random.default_rng(0)
order = rng.permutation(n)
train_genes, val_genes = [genes[i] for i in order[: 3 * n // 4]], [genes[i] for i in order[3 * n // 4:]]
for label in ("first open (page cache warm from the write)", "second open"):
    t0 = time.perf_counter()
    ps = pack.PackedStore(path)
    tr = ps.store(train_genes, device=dev)
    va = ps.store(val_genes, device=dev)
    torch.cuda.synchronize()
    print("%s: %d + %d genes resident in %.2f s" % (label, len(tr), len(va), time.perf_counter() - t0))
    del tr, va
os.remove(path)
