"""Soak / determinism check of the shipped single-GPU training loop: the same N steps of `train_epoch` (graph replay or eager, fused optimiser, riders, head
ride, in-launch batch gather and step log) twice from the same seed -- parameters, moments and the logged losses must agree bit for bit.  A race that
loses one arrival in ten thousand shows up here; the GPU tests run too few steps for it.
    python tools/soak_determinism.py [steps=6000] [graph|eager]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import EpochFeed, Trainer
from chromoformer_amd.synth import synthetic_store
from chromoformer_amd.train import epoch_permutation, train_epoch
from chromoformer_amd.data import shard_indices

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
graph = not (len(sys.argv) > 2 and sys.argv[2] == "eager")
B = 64
dev = torch.device("cuda", 0)
store = synthetic_store(16384, dev, seed=4321)


def run():
    model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
    trainer = Trainer(model, lr=3e-5, use_graph=graph)
    feed = EpochFeed(model, store, B)
    losses = []
    done = 0
    torch.manual_seed(1234)                      # (epoch_permutation draws from the global RNG, as the DataLoader does)
    t0 = time.perf_counter()
    while done < steps:
        batches = shard_indices(epoch_permutation(len(store)), 0, 1, B)[: steps - done]
        train_epoch(trainer, feed, batches, lambda lo, la, ls: losses.append(ls.numpy().copy()))
        done += len(batches)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return model._flat.cpu().clone(), model._mflat.cpu().clone(), model._vflat.cpu().clone(), np.concatenate(losses) if losses else np.zeros(0), dt


a = run()
b = run()
same = all(torch.equal(x, y) for x, y in zip(a[:3], b[:3])) and np.array_equal(a[3], b[3])
print("soak %s: %d steps twice, %.3f / %.3f ms per step, %d logged losses, finite %s, bit-identical %s" % (
    "graph" if graph else "eager", steps, 1e3 * a[4] / steps, 1e3 * b[4] / steps, a[3].size, bool(np.isfinite(a[3]).all() and torch.isfinite(a[0]).all()), same))
sys.exit(0 if same else 1)
