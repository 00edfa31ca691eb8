"""Phase timeline of the fused centre-row trunk kernels (workgroup of gene 0 at the longest resolution), in s_memtime ticks and as a
share of the workgroup's time (the tick rate is not documented for gfx950; against the kernel durations of rocprofv3 it is ~2.4 GHz
here):   python tools/trunk_stamps.py"""
import os, sys
os.environ["CF_STAMP_TRUNK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.synth import synthetic_batch
B = 64
m = ChromoformerClassifier(max_batch=B).cuda(0)
packed = m.pack_batch(synthetic_batch(B, seed=1, regime=os.environ.get("REGIME", "dense")))
for _ in range(3):
    m.forward_backward(packed, torch.zeros(B, dtype=torch.long))
torch.cuda.synchronize()
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64).astype(np.int64)
fwd = ["x0 + q chain E", "attention E", "post chain E (+lin_proj_p)", "q chain P0", "attention P0", "post chain P0 (+q chain P1)", "attention P1", "post chain P1"]
bwd = ["post chain P1", "attention P1", "q chain P1", "post chain P0", "attention P0", "q chain P0", "join + lin_proj_p", "post chain E", "attention E",
       "q chain E", "7-mark partials"]
for name, base, names in (("forward", 0, fwd), ("backward", 32, bwd)):
    d = np.diff(t[base: base + len(names) + 1])
    print("%s: total %d ticks" % (name, d.sum()))
    for n, v in zip(names, d):
        print("   %-32s %7d ticks  %5.1f %%" % (n, v, 100.0 * v / d.sum()))
