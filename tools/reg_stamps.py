"""Phase timeline of the fused Regulation forward kernel (workgroup 0): CF_STAMP=1 python tools/reg_stamps.py"""
import os, sys
BWD = len(sys.argv) > 1 and sys.argv[1] == "bwd"
os.environ["CF_STAMP_BWD" if BWD else "CF_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.synth import synthetic_batch
B = 64
m = ChromoformerClassifier(max_batch=B).cuda(0)
packed = m.pack_batch(synthetic_batch(B, seed=1, regime="dense"))
for _ in range(3):
    m.forward_backward(packed, torch.zeros(B, dtype=torch.long))
torch.cuda.synchronize()
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64)[: 16 * 6].reshape(6, 16).astype(np.int64)
import ctypes
if os.environ.get("CF_REG8", "1") != "0":      # 512-thread kernels (cf_reg8.h)
    if BWD:
        names = ["start", "ln2 bwd", "dpre1 (W2)", "dy1 (W1)", "ln1 bwd", "da (Wo) + attention", "dgrad K=1024"]
    else:
        names = ["start", "q|k|v|g", "attention", "barrier", "Wo+res", "LN1", "W1", "W2", "LN2"]
    n = len(names)
    for l in range(6):
        d = np.diff(t[l, :n])
        print("layer %d total %6d cyc: " % (l, t[l, n - 1] - t[l, 0]) + "  ".join("%s %d" % (names[i + 1], d[i]) for i in range(n - 1)))
    sys.exit(0)
if BWD:
    names = ["start", "ln2 prep", "ln2 bwd", "dpre1 (W2)", "dy1 (W1)", "ln1 bwd", "da (Wo)+loads", "barrier", "gate/do", "dp", "softmax bwd", "dq dk dv", "dgrad K=1024"]
    for l in range(6):
        d = np.diff(t[l, :13])
        print("layer %d total %6d cyc: " % (l, t[l, 12] - t[l, 0]) + "  ".join("%s %d" % (names[i + 1], d[i]) for i in range(12)))
    sys.exit(0)
names = ["start", "qkvg done", "barrier", "scores", "softmax", "gate*pv", "barrier", "Wo+res", "LN1", "W1", "W2", "LN2"]
for l in range(6):
    d = np.diff(t[l, :12])
    print("layer %d total %6d cyc: " % (l, t[l, 11] - t[l, 0]) + "  ".join("%s %d" % (names[i + 1], d[i]) for i in range(11)))
