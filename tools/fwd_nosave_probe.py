"""Forward pass with and without the activation saves (k_reg8_fwd<.., false> vs <.., true> under rocprofv3 --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import Trainer
from chromoformer_amd.synth import synthetic_batch
B = 64
m = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
tr = Trainer(m, use_graph=False)
slot = tr.stage(synthetic_batch(B, seed=1, regime="dense"))
for _ in range(50):
    tr.evaluate(slot)          # save = 0
for _ in range(50):
    tr.step(slot)              # save = 1
torch.cuda.synchronize()
