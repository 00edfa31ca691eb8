"""Timeline of k_head_train (workgroup 0, thread 0), s_memtime ticks:   python tools/head_stamps.py"""
import os, sys
os.environ["CF_STAMP_HEAD"] = "1"
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
t = m.debug_buffer("reg_tdbg").cpu().numpy().view(np.uint64).astype(np.int64)[128:136]
names = ["requests issued -> inputs in LDS (barrier)", "hidden layer (K = 384 product)", "logits", "loss + d logits", "dh1", "dhin product + stores", "loss reduction (fences)"]
print("total %d ticks" % (t[7] - t[0]))
for n, a, b in zip(names, t[:-1], t[1:]):
    print("   %-46s %6d ticks" % (n, b - a))
