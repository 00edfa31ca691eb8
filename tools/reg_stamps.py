"""Phase timeline of the fused Regulation kernels, workgroup (gene 0, resolution 0), every wave:
   python tools/reg_stamps.py [fwd|bwd|nosave] [layer]      (stamps: CF_STAMP8 in csrc/cf_reg8.h; shader-clock ticks)
Each line is one wave: ticks from the layer's first stamp of wave 0 to the wave's stamp at the END of each phase.  A stamp
in front of a barrier is the wave's arrival, the one behind it the release: waves 0 and 4 (1 and 5, ...) share a SIMD."""
import os, sys
BWD = len(sys.argv) > 1 and sys.argv[1] == "bwd"
NOSAVE = len(sys.argv) > 1 and sys.argv[1] == "nosave"      # the inference instantiation of the forward (no activation saves)
LAYER = int(sys.argv[2]) if len(sys.argv) > 2 else 2
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
if NOSAVE:
    for _ in range(2):
        m._run_forward(packed[0], save=False)
torch.cuda.synchronize()
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64)[: 8 * 6 * 16].reshape(8, 6, 16).astype(np.int64)
if BWD:
    names = ["start", "ln2 bwd+req |", "| dpre1 (W2) |", "| dy1 (W1) |", "| ln1 bwd+req |", "| da (Wo) + attention |", "| dgrad K=1024"]
    order = range(5, -1, -1)
else:
    names = ["start", "q|k|v|g", "attention", "gate.o |", "| Wo+res |", "| LN1 |", "| W1 |", "| W2 |", "| LN2"]
    order = range(6)
n = len(names)
print("layer totals (wave 0):", "  ".join("%d" % (t[0, l, n - 1] - t[0, l, 0]) for l in order))
print("layer %d, ticks since wave 0 entered the layer;  columns: %s" % (LAYER, "  ".join(names[1:])))
t0 = t[0, LAYER, 0]
for w in range(8):
    extra = ""
    if not BWD and t[w, LAYER, 9] > t0:      # built with -DCF_STAMP_CHUNKS: ends of the q, k, v chunks of the projection
        extra = "   chunks q/k/v: " + " ".join("%6d" % (t[w, LAYER, i] - t0) for i in (9, 10, 11))
    print("wave %d: start %5d  " % (w, t[w, LAYER, 0] - t0) + "  ".join("%6d" % (t[w, LAYER, i] - t0) for i in range(1, n)) + extra)
