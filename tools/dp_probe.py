"""The data-parallel step on a one-rank RCCL group, for a kernel trace (rocprofv3 --kernel-trace -- python3 tools/dp_probe.py [graph|eager] [steps]);
tools/timeline.py <trace.csv> <step> k_trunk_fwd  then prints one step's kernels with start offsets (gaps = host / event / stream-sync overhead)."""
import os, socket, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from chromoformer_amd import ChromoformerClassifier
from chromoformer_amd.engine import Trainer
from chromoformer_amd.synth import synthetic_batch
graph = len(sys.argv) > 1 and sys.argv[1] == "graph"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=dev)
model = ChromoformerClassifier(seed=42, max_batch=64).cuda(0)
tr = Trainer(model, lr=3e-5, world_size=1, process_group=dist.group.WORLD, use_graph=graph)
slot = tr.stage(synthetic_batch(64, seed=1234, regime="dense"))
import time
for _ in range(30):
    tr.step(slot)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.step(slot)
torch.cuda.synchronize()
print("dp %s: %.4f ms per step" % ("graph" if graph else "eager", 1e3 * (time.perf_counter() - t0) / steps))
dist.destroy_process_group()
