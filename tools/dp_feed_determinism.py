"""Run-to-run determinism of the data-parallel step under a feed, two ranks on ONE device over gloo (the test configuration; RCCL refuses two
ranks per device): per-step bit checksums of every gradient bucket and of the parameters, taken on the trainer's stream without host
synchronisation, so that two runs can be compared step by step.  Round 5 used it to find that a gene's prediction head was skipped in about one
run of ten (stale logits, every gradient of the step wrong) -- the head-ride arrival counters were zeroed by a workgroup at launch start and
rewound by the last arriver; since then they are monotonic (csrc/cf_head_ride.h).
   python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dp_feed_determinism.py out.pt 7
   (repeat, then compare the "sums" / "losses" of the saved files; DBG_EVAL=0, DBG_LAST=n, DP_GRAPH=0, CF_DP_HALVES=0 vary the run)"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import Trainer, EpochFeed
from chromoformer_amd.synth import synthetic_store
from chromoformer_amd.data import shard_indices
out, steps = sys.argv[1], int(sys.argv[2])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
model = ChromoformerClassifier(seed=42, max_batch=4).cuda(0)
store = synthetic_store(64, dev, seed=5, regime="realistic")
tr = Trainer(model, lr=3e-5, world_size=world, process_group=dist.group.WORLD, use_graph=(os.environ.get("DP_GRAPH", "1") == "1"))
feed = EpochFeed(model, store, 4)
perm = list(range(64))
batches = shard_indices(perm, rank, world, 4, drop_last=True)[:steps]
val = synthetic_store(6, dev, seed=9, regime="realistic")
for ep in range(2):
    nb = batches if ep == 0 else batches[:int(os.environ.get("DBG_LAST", "1"))]
    feed.begin_epoch(nb, tr.stream)
    for k in range(len(nb)):
        tr.step(feed.slot)
        if os.environ.get("DBG_SUMS", "1") == "1":      # per-step checksums computed ON the trainer's stream: no host synchronisation
            from chromoformer_amd import _lib
            with torch.cuda.stream(tr.stream):
                row = []
                for b in (_lib.BUCKET_PE, _lib.BUCKET_REG_LO, _lib.BUCKET_REG_HI):
                    g = tr._buckets.get(b)
                    if g is None:
                        off, n = (0, tr._buckets[_lib.BUCKET_PE].numel()) if b == _lib.BUCKET_PE else (0, 0)
                        continue
                    off = (g.data_ptr() - model._gflat.data_ptr()) // 4
                    pw = model._flat[off: off + g.numel()]
                    row += [g.view(torch.int32).to(torch.int64).sum(), pw.view(torch.int32).to(torch.int64).sum()]
                globals().setdefault("sums", []).append(torch.stack(row))
    torch.cuda.synchronize()
    losses = globals().setdefault("losses", [])
    losses.append(feed.loss_log[:len(nb)].clone())
    if os.environ.get("DBG_EVAL", "1") == "1":
        ev = tr.evaluate_store(val, 4)
        losses.append(ev.detach().cpu().reshape(-1).clone())
    tr.scheduler_step()
torch.cuda.synchronize()
if rank == 0:
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    g = model._gflat.detach().cpu()
    for e in model._table:
        if e["trainable"]:
            sd["grad:" + e["name"]] = g[e["offset"]:e["offset"] + e["numel"]].clone()
    sd["losses"] = torch.cat(losses)
    sd["sums"] = torch.stack(globals()["sums"]).cpu() if "sums" in globals() else torch.zeros(1, 1)
    torch.save(sd, out)
dist.barrier()
dist.destroy_process_group()
