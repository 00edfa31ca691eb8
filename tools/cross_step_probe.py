"""Timing probe (WRONG dependencies, time only): the single-GPU step with the non-rider Regulation tiles of step k (reduction + AdamW) on a
low-priority side stream, beside the NEXT step's trunk forward, instead of in the reduction launch at the end of step k.
    python tools/cross_step_probe.py [steps]
Upper bound of what a cross-step schedule could win: no join in front of the Regulation forward, the re-tiling riders race with the updates."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chromoformer_amd import ChromoformerClassifier, _lib
from chromoformer_amd.engine import Trainer
from chromoformer_amd.synth import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 64
model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
batch = synthetic_batch(B, seed=1234, regime="dense")
L = _lib.lib()


def run(mode, prio):
    tr = Trainer(model, lr=3e-5, use_graph=False)
    if prio:
        tr.stream = torch.cuda.Stream(device=model._device, priority=-1)
        tr.side = torch.cuda.Stream(device=model._device, priority=0)
    slot = tr.stage(batch)
    ev = torch.cuda.Event()
    h = model._handle

    def step():
        with torch.cuda.stream(tr.stream):
            st, side = tr.stream.cuda_stream, tr.side.cuda_stream
            model._sync_tiled(st)
            tr._seq_early(slot, st, reduce=False, gather=True)
            model._step += 1
            hp = (tr.lr, 0.9, 0.999, 1e-8, 0.01, model._step)
            _lib.check(L.cf_rider_arm(h, *hp, 0, 512), "arm")
            tr._part(slot, st, 4)
            if mode == "split_main":          # two launches on the main stream (what the split alone costs)
                _lib.check(L.cf_reduce_opt_part(h, B, _lib.BUCKET_REG, *hp, 0, st), "reg")
                _lib.check(L.cf_reduce_opt_part(h, B, _lib.BUCKET_PE, *hp, 0, st), "pe")
            else:                              # the Regulation remainder on the side stream, nobody waits for it
                ev.record(tr.stream)
                tr.side.wait_event(ev)
                _lib.check(L.cf_reduce_opt_part(h, B, _lib.BUCKET_REG, *hp, 0, side), "reg")
                _lib.check(L.cf_reduce_opt_part(h, B, _lib.BUCKET_PE, *hp, 0, st), "pe")

    for _ in range(50):
        step() if mode != "base" else tr.step(slot)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step() if mode != "base" else tr.step(slot)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / steps


for mode, prio in (("base", False), ("split_main", False), ("side", False), ("side", True), ("base", False), ("side", True)):
    print("%-12s priority streams %-5s  %.4f ms per step" % (mode, prio, run(mode, prio)), flush=True)
