"""What a polled event costs the training loop: steps alone, + record_event every 10, + query, + window/report."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import EpochFeed, Trainer
from chromoformer_amd.synth import synthetic_store
from chromoformer_amd.train import epoch_permutation
from chromoformer_amd.data import shard_indices
B = 64
dev = torch.device("cuda", 0)
model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
store = synthetic_store(16384, dev, seed=4321)
trainer = Trainer(model, lr=3e-5)
feed = EpochFeed(model, store, B)
batches = shard_indices(epoch_permutation(len(store)), 0, 1, B)
N = 250
from chromoformer_amd.train import _report_train
quiet = lambda *a, **k: None
wb = type("W", (), {"log": staticmethod(quiet)})
report = lambda lo, la, ls: _report_train(quiet, wb, 1, float(ls.numpy().mean()), trainer.lr, lo, la, False)
def run(mode):
    feed.begin_epoch(batches[:N], trainer.stream)
    torch.cuda.synchronize()
    pending = []
    t0 = time.perf_counter()
    for k in range(1, N + 1):
        trainer.step(feed.slot)
        if mode >= 1 and k % 10 == 0:
            ev = torch.cuda.Event() if mode < 4 else torch.cuda.Event(blocking=False, enable_timing=False)
            ev.record(trainer.stream)
            pending.append(ev)
            if mode >= 2:
                while pending and pending[0].query():
                    pending.pop(0)
                    if mode >= 3:
                        feed.window(0, 10)
    t1 = time.perf_counter()
    if mode == 5:
        for ev in pending:
            ev.synchronize()
    if mode in (8, 9):
        for i, ev in enumerate(pending):
            ev.synchronize()
            w_ = feed.window(i * 10, i * 10 + 10)
            if mode == 9:
                report(*w_)
    if mode == 6:
        for ev in pending:
            while not ev.query():
                time.sleep(0.0002)
    if mode == 7:
        for ev in pending:
            while not ev.query():
                pass
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t0) / N
for rep_ in range(2):
    if rep_ == 1:
        torch.set_num_threads(1)
        print("torch.set_num_threads(1)")
    for mode, name in ((0, "steps only"), (1, "+ record every 10"), (2, "+ query"), (3, "+ window"), (5, "drain: synchronize"), (6, "drain: query + sleep"), (7, "drain: query spin"), (8, "drain: sync + window"), (9, "drain: sync + window + report"), (0, "steps only")):
        h, ms = run(mode)
        print("%-20s host loop %7.1f ms   %.4f ms/step" % (name, h, ms))
