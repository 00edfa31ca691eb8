"""Step time of configurations away from the default shape (they run the stand-alone kernels layer by layer): one training step at bsz 64,
dense synthetic batch, eager and as a replayed graph.   python tools/variant_step_time.py [d_emb ...]      (default: 128 64 256)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import Trainer
from chromoformer_amd.synth import synthetic_batch

for d in [int(x) for x in sys.argv[1:]] or [128, 64, 256]:
    embed = dict(n_layers=1, n_heads=2, d_model=d, d_ff=128)
    pair = dict(n_layers=2, n_heads=2, d_model=d, d_ff=256)
    reg = dict(n_layers=6, n_heads=8, d_model=256, d_ff=256)
    model = ChromoformerClassifier(7, d, 128, embed, pair, reg, seed=42, max_batch=64).cuda(0)
    out = []
    for graph in (False, True):
        tr = Trainer(model, lr=3e-5, use_graph=graph)
        slot = tr.stage(synthetic_batch(64, seed=1234, regime="dense"))
        for _ in range(30):
            tr.step(slot)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            tr.step(slot)
        torch.cuda.synchronize()
        out.append(1e3 * (time.perf_counter() - t0) / 200)
    print("d_emb %3d: %.4f ms per step eager, %.4f replayed  (%.1f k genes/s)" % (d, out[0], out[1], 64 / min(out)), flush=True)
    del model
