"""Step time of the engine in its three launch modes (one hipGraph; graph split around the timed kernel; eager)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import Trainer
from chromoformer_amd.synth import synthetic_batch
B = 64
batch = synthetic_batch(B, seed=1234, regime="dense")
for name, kw in (("one graph", dict(use_graph=True)), ("split graph (k_reg_bwd timed)", dict(use_graph=True, timed_kernel="k_reg_bwd")),
                 ("eager", dict(use_graph=False))):
    res = []
    for rep in range(2):
        m = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
        tr = Trainer(m, **kw)
        slot = tr.stage(batch)
        for _ in range(20):
            tr.step(slot)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(400):
            tr.step(slot)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 400 * 1e3)
    print("%-32s %.4f %.4f ms/step" % (name, *res))
