"""Worker of tests/test_dp_gpu.py: one data-parallel rank of the step engine (launched by torch.distributed.run).
argv: <out.pt> <steps>.  Every rank drives device 0 over gloo (RCCL refuses two ranks on one device); rank r trains
on slice r of each global batch of 8 genes and rank 0 saves the final parameters."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chromoformer_amd import ChromoformerClassifier  # noqa: E402
from chromoformer_amd.data import shard_indices  # noqa: E402
from chromoformer_amd.engine import Trainer  # noqa: E402
from oracle import chromoformer_oracle as orc  # noqa: E402
from tests.helpers import take  # noqa: E402

out, steps = sys.argv[1], int(sys.argv[2])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.cuda.set_device(0)
G = 8
model = ChromoformerClassifier(seed=42, max_batch=G // world).cuda(0)
trainer = Trainer(model, lr=3e-5, world_size=world, process_group=dist.group.WORLD, use_graph=(os.environ.get("DP_GRAPH", "1") == "1"))
losses = []
for k in range(steps):
    batch = orc.synthetic_batch(G, seed=100 + k, regime="realistic")
    idx = shard_indices(list(range(G)), rank, world, G // world)[0]
    slot = trainer.stage(take(batch, idx))
    _, loss = trainer.step(slot)
    with torch.cuda.stream(trainer.stream):
        losses.append(loss.clone())
torch.cuda.synchronize()
mean_loss = torch.stack([x.reshape(()) for x in losses]).cpu()
dist.all_reduce(mean_loss)                       # per-rank losses are means over the shard scaled by nothing: average them
if rank == 0:
    torch.save({"net": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "loss": mean_loss / world}, out)
dist.barrier()
dist.destroy_process_group()
