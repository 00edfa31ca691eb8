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
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64)[256:256 + 32].astype(np.int64)
names = ["stage", "u", "t=vin.PE^T + epilogue", "softmax", "w", "sum p PE (mfma)", "reduce+epilogue"]
for k, nm in ((0, "fwd"), (16, "bwd")):
    d = np.diff(t[k:k + 8])
    print(nm, "total", t[k + 7] - t[k], " ".join("%s %d |" % (names[i], d[i]) for i in range(7)))
