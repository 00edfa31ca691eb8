"""Where the wall time of one training epoch goes on the host side (train.train_epoch over a resident synthetic split):
    python tools/epoch_probe.py [n_genes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import EpochFeed, Trainer
from chromoformer_amd.synth import synthetic_store
from chromoformer_amd.train import _report_train, epoch_permutation, train_epoch
from chromoformer_amd.data import shard_indices
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14217
dev = torch.device("cuda", 0)
model = ChromoformerClassifier(seed=42, max_batch=64).cuda(0)
store = synthetic_store(n, dev, seed=1, regime="realistic")
trainer = Trainer(model, lr=3e-5)
feed = EpochFeed(model, store, 64)
lines = []
say = lambda *a, **k: lines.append(a)
wb = type("W", (), {"log": staticmethod(lambda *a, **k: None)})
report = lambda lo, la, ls: _report_train(say, wb, 1, float(ls.numpy().mean()), trainer.lr, lo, la, False)
for ep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    perm = epoch_permutation(n)
    t1 = time.perf_counter()
    batches = shard_indices(perm, 0, 1, 64, drop_last=True)
    t2 = time.perf_counter()
    feed.begin_epoch(batches, trainer.stream)
    t3 = time.perf_counter()
    pending = []
    for k in range(1, len(batches) + 1):
        trainer.step(feed.slot)
        if k % 10 == 0:
            pending.append((k - 10, k, trainer.stream.record_event()))
            while pending and pending[0][2].query():
                lo, hi, _ = pending.pop(0)
                report(*feed.window(lo, hi))
    t4 = time.perf_counter()
    for lo, hi, ev in pending:
        ev.synchronize()
        report(*feed.window(lo, hi))
    t5 = time.perf_counter()
    torch.cuda.synchronize()
    t6 = time.perf_counter()
    nb = len(batches)
    print("epoch %d: %d steps  permutation %.1f ms  shard_indices %.1f  begin_epoch %.1f  issue loop %.1f (%.4f ms/step)  drain + late reports %.1f (%d windows)  final sync %.1f  total %.1f ms = %.4f ms/step"
          % (ep, nb, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t3) / nb, 1e3 * (t5 - t4), len(pending), 1e3 * (t6 - t5), 1e3 * (t6 - t0), 1e3 * (t6 - t0) / nb))
