"""Timeline of the rider waves of k_trunk_bwd (workgroup 0 of the rider row) against the trunk workgroup of gene 0 at the longest
resolution, in s_memtime ticks:   python tools/rider_stamps.py [rider_tiles]"""
import os, sys
os.environ["CF_STAMP_TRUNK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import Trainer
from chromoformer_amd.synth import synthetic_batch
B = 64
m = ChromoformerClassifier(max_batch=B).cuda(0)
tr = Trainer(m, use_graph=False, rider_tiles=int(sys.argv[1]) if len(sys.argv) > 1 else None)
slot = tr.stage(synthetic_batch(B, seed=1, regime="dense"))
for _ in range(4):
    tr.step(slot)
torch.cuda.synchronize()
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64).astype(np.int64)
t0 = t[32]
print("trunk workgroup (gene 0, longest resolution): %d ticks" % (t[32 + 11] - t0))
print("trunk workgroups of gene 0, longest resolution first: " + "  ".join("%d ticks" % v for v in t[100:103]))
for w in range(8):
    s = t[64 + 4 * w: 64 + 4 * w + 4]
    print("rider wave %d: starts %6d   stage loop %6d .. %6d (%6d)   optimiser epilogue done %6d (%6d)" % (w, s[3] - t0, s[0] - t0, s[1] - t0, s[1] - s[0], s[2] - t0, s[2] - s[1]))
