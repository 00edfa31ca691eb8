"""Phase timeline of the team forward of the Regulation stack (cf_regq.h), member 0 of unit 0, wave 0:
   python tools/regq_stamps.py      (stamps: CF_STAMPQ; shader-clock ticks)"""
import os, sys
os.environ["CF_STAMP"] = "1"
os.environ["CF_REG_TEAM"] = "1"
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
raw = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64).astype(np.int64)
t = raw[: 6 * 16].reshape(6, 16)
names = ["projection (2 chunks)", "Wo operands requested / softmax", "-> barrier", "p v, gate", "-> barrier", "Wo partial stored, W1 requested", "arrive + wait 1",
         "sum 1", "LN1", "W1 (+W2 requested)", "W2 partial stored, next ring requested", "arrive + wait 2", "sum 2", "LN2"]
print("layer totals:", " ".join("%d" % (t[l, 13] - t[l, 0]) for l in range(6)), " whole stack:", t[5, 13] - t[0, 0])
for l in (1, 3):
    print("layer %d:" % l, " | ".join("%s %d" % (names[i], t[l, i] - t[l, i - 1] if i else 0) for i in range(1, 14)))
NW = 464      # (the debug buffer holds the first 464 workgroups)
se = raw[96: 96 + 2 * NW].reshape(NW, 2)
t0 = se[:, 0].min()
dur = se[:, 1] - se[:, 0]
print("workgroups: first start 0, last start %d, first end %d, last end %d; duration min / median / max %d / %d / %d" % (
    se[:, 0].max() - t0, se[:, 1].min() - t0, se[:, 1].max() - t0, dur.min(), int(np.median(dur)), dur.max()))
for x in range(8):
    sel = se[np.arange(NW) % 8 == x]
    print("  XCD %d: starts %d..%d ends %d..%d" % (x, sel[:, 0].min() - t0, sel[:, 0].max() - t0, sel[:, 1].min() - t0, sel[:, 1].max() - t0))
