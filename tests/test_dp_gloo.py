"""Data-parallel path on CPU (gloo, world_size 2): rank-sharded batches + per-rank mean-loss
gradients scaled by 1/world + all-reduce(SUM) over the flat active-gradient range reproduce the
single-process gradient of the global batch.  The gradient payload comes from the CPU oracle; the
flat layout is the product's (cf_param_layout, host only)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from chromoformer_amd import _lib
from chromoformer_amd.data import shard_indices
from oracle import chromoformer_oracle as orc
from oracle import restructured as rst
from tests.helpers import take
from tests.test_abi_cpu import _cfg


def _flat_grads(batch, label, scale):
    lay, tab = _lib.param_layout(_cfg())
    P = orc.init_params(None, 42, False)
    for t in P.values():
        t.requires_grad_(True)
    loss = orc.loss_fn(rst.forward(P, batch), label) * scale
    loss.backward()
    flat = torch.zeros(lay.n_total)
    for e in tab:
        if e["trainable"]:
            flat[e["offset"]:e["offset"] + e["numel"]] = P[e["name"]].grad.reshape(-1)
    return flat, lay.n_active


def _worker(rank, world, port, B, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    batch = orc.synthetic_batch(B, seed=77, regime="realistic")
    idx = shard_indices(list(range(B)), rank, world, B // world)[0]
    sub = take(batch, idx)
    flat, n_active = _flat_grads(sub, sub["label"], 1.0 / world)
    active = flat[:n_active]
    dist.all_reduce(active)                       # SUM, in place on the contiguous active range
    if rank == 0:
        ret["flat"] = flat.clone()
    dist.destroy_process_group()


def test_two_rank_allreduce_equals_global_batch_gradient():
    B, world = 4, 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, B, ret), nprocs=world, join=True)
    batch = orc.synthetic_batch(B, seed=77, regime="realistic")
    ref, n_active = _flat_grads(batch, batch["label"], 1.0)
    got = ret["flat"]
    assert (got[n_active:] == 0).all()
    err = (got - ref).abs().max().item()
    assert err <= 1e-5 * ref.abs().max().item() + 1e-9, err
