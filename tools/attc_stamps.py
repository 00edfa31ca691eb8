"""Phase timeline of k_attc2 (workgroup 0, L = 400): python tools/attc_stamps.py"""
import os, sys
os.environ["CF_STAMP_ATTC"] = "1"
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
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64)[256:256 + 256].astype(np.int64).reshape(8, 32)
names = ["stage", "products + u", "epilogue + block statistics", "softmax combine", "w", "sum p PE (mfma)", "reduce+epilogue"]
for k, nm in ((0, "fwd"), (16, "bwd")):
    t0 = t[0, k]
    d = np.diff(t[0, k:k + 8])
    print(nm, "wave 0: total", t[0, k + 7] - t0, " ".join("%s %d |" % (names[i], d[i]) for i in range(7)))
    for wv in range(8):      # every wave, relative to wave 0's first stamp: start, before barrier 1, after, products done, before barrier 2, after, ...
        print("   wave %d: start %6d | b1 %6d -> %6d | products done %6d | b2 %6d -> %6d | b3 -> %6d | b4 -> %6d | pass 5 done %6d | end %6d" % (
            (wv,) + tuple(t[wv, k + i] - t0 for i in (0, 10, 1, 8, 9, 2, 3, 4, 6, 7))))
