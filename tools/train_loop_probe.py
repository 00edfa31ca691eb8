"""Where the time of the shipped training loop goes: steps with / without the metric windows, and the kernels' own time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import EpochFeed, Trainer
from chromoformer_amd.synth import synthetic_store
from chromoformer_amd.train import _report_train, epoch_permutation, train_epoch
from chromoformer_amd.data import shard_indices
B = 64
dev = torch.device("cuda", 0)
model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
store = synthetic_store(int(os.environ.get("GENES", "16384")), dev, seed=4321)
trainer = Trainer(model, lr=3e-5)
feed = EpochFeed(model, store, B)
quiet = lambda *a, **k: None
wb = type("W", (), {"log": staticmethod(quiet)})
report = lambda lo, la, ls: _report_train(quiet, wb, 1, float(ls.numpy().mean()), trainer.lr, lo, la, False)
batches = shard_indices(epoch_permutation(len(store)), 0, 1, B)
train_epoch(trainer, feed, batches[:30], report)
torch.cuda.synchronize()
for name, rep in (("no report", None), ("report", report), ("no report", None)):
    t0 = time.perf_counter()
    train_epoch(trainer, feed, batches[:200], rep)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-10s 200 steps: host loop %.1f ms, total %.1f ms -> %.3f ms/step" % (name, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t2 - t0) / 200))

# ---- cost of one metric window, piece by piece
torch.cuda.synchronize()
import numpy as np
def timeit(f, n=20):
    f()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    return 1e3 * (time.perf_counter() - t0) / n
w = feed.window(0, 10)
print("window() with idle GPU      %.3f ms" % timeit(lambda: feed.window(0, 10)))
print("_report_train               %.3f ms" % timeit(lambda: report(*w)))
print("softmax+argmax (torch cpu)  %.3f ms" % timeit(lambda: (w[0].softmax(axis=1)[:, 1].numpy(), w[0].argmax(axis=1).numpy())))
from chromoformer_amd.train import binary_auc_ap
sc, la = w[0].softmax(axis=1)[:, 1].numpy(), w[1].numpy()
print("binary_auc_ap               %.3f ms" % timeit(lambda: binary_auc_ap(la, sc)))
# window() while the GPU is busy with queued steps
feed.begin_epoch(batches[:200], trainer.stream)
for _ in range(100):
    trainer.step(feed.slot)
t0 = time.perf_counter(); feed.window(0, 10); t1 = time.perf_counter()
print("window() with 100 steps queued: %.3f ms" % (1e3 * (t1 - t0)))
torch.cuda.synchronize()

# ---- the loop of train_epoch, instrumented
feed.begin_epoch(batches[:200], trainer.stream)
pending, t_step, t_rec, t_sync, t_rep = [], 0.0, 0.0, 0.0, 0.0
T0 = time.perf_counter()
for k in range(1, 201):
    t0 = time.perf_counter(); trainer.step(feed.slot); t_step += time.perf_counter() - t0
    if k % 10 == 0:
        t0 = time.perf_counter(); pending.append((k - 10, k, trainer.stream.record_event())); t_rec += time.perf_counter() - t0
        if len(pending) > 1:
            lo, hi, ev = pending.pop(0)
            t0 = time.perf_counter(); ev.synchronize(); t_sync += time.perf_counter() - t0
            t0 = time.perf_counter(); report(*feed.window(lo, hi)); t_rep += time.perf_counter() - t0
torch.cuda.synchronize()
print("instrumented: total %.1f ms; step calls %.1f, record_event %.1f, ev.synchronize %.1f, window+report %.1f" %
      (1e3 * (time.perf_counter() - T0), 1e3 * t_step, 1e3 * t_rec, 1e3 * t_sync, 1e3 * t_rep))
