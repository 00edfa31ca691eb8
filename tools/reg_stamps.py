"""Phase timeline of the fused Regulation forward kernel (workgroup 0): CF_STAMP=1 python tools/reg_stamps.py"""
import os, sys
os.environ["CF_STAMP"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chromoformer_amd import ChromoformerClassifier
from oracle import chromoformer_oracle as orc
B = 64
m = ChromoformerClassifier(max_batch=B).cuda(0)
packed = m.pack_batch(orc.synthetic_batch(B, seed=1, regime="dense"))
for _ in range(3):
    m.forward_backward(packed, torch.zeros(B, dtype=torch.long))
torch.cuda.synchronize()
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64)[: 16 * 6].reshape(6, 16).astype(np.int64)
names = ["start", "qkvg done", "barrier", "scores", "softmax", "gate*pv", "barrier", "Wo+res", "LN1", "W1", "W2", "LN2"]
for l in range(6):
    d = np.diff(t[l, :12])
    print("layer %d total %6d cyc: " % (l, t[l, 11] - t[l, 0]) + "  ".join("%s %d" % (names[i + 1], d[i]) for i in range(11)))
