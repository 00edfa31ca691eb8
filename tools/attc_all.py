"""(start, end) of every k_attc2 workgroup of the last launch: CF_STAMP_ATTC_ALL=0 (forward) or 1 (backward)  python tools/attc_all.py"""
import os, sys
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
ag = int(os.environ.get("CF_STAMP_ATTC_AG", "8"))
nx = (512 if ag == 8 else 64) // ag
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64)[256:256 + 3 * nx * 2].astype(np.int64).reshape(3, nx, 2)
t0 = t[..., 0].min()
for y in range(3):
    st, en = t[y, :, 0] - t0, t[y, :, 1] - t0
    print("blockIdx.y %d (L = %s): start min %d max %d | end min %d median %d max %d | duration min %d median %d max %d" % (
        y, ("400", "80", "20")[y], st.min(), st.max(), en.min(), int(np.median(en)), en.max(), (en - st).min(), int(np.median(en - st)), (en - st).max()))
    if y == 0:
        print("   durations by workgroup:", " ".join(str(int(d)) for d in (en - st)))
# shader clocks differ between clock domains (XCDs): cluster the workgroups by start value, then look inside a cluster
flat = t.reshape(-1, 2)
order = np.argsort(flat[:, 0])
groups, cur = [], [order[0]]
for a_, b_ in zip(order[:-1], order[1:]):
    if flat[b_, 0] - flat[a_, 0] > 200000:
        groups.append(cur)
        cur = []
    cur.append(b_)
groups.append(cur)
for g in groups:
    st, en = flat[g, 0], flat[g, 1]
    print("clock domain with %2d workgroups (linear ids mod 8: %s): first start -> last end %6d | start skew max %5d | longest workgroup %6d" % (
        len(g), sorted(set(int(i) % 8 for i in g)), en.max() - st.min(), st.max() - st.min(), (en - st).max()))
