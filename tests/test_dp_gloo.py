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


def _shard_worker(rank, world, port, store_path, genes, bsz, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from chromoformer_amd import pack
    from chromoformer_amd.data import static_epoch_batches, static_shard
    mine = static_shard(genes, rank, world)
    store = pack.PackedStore(store_path).store(mine)                 # a rank loads its shard only
    g = torch.Generator().manual_seed(5)
    perm = torch.randperm(len(genes), generator=g).tolist()          # the same draws on every rank
    batches = static_epoch_batches(perm, rank, world, bsz)
    sizes = [None] * world
    dist.all_gather_object(sizes, (len(store), len(batches)))
    named = [[mine[i] for i in b] for b in batches]
    allb = [None] * world
    dist.all_gather_object(allb, named)
    if rank == 0:
        ret["sizes"], ret["batches"] = sizes, allb
    dist.destroy_process_group()


def test_static_sharding_loads_one_worlds_th_of_the_split_per_rank(tmp_path):
    """train.py --dp-shard static: every rank opens the packed store and gathers ONLY its genes; the epoch's global batches
    are disjoint unions of one equal-sized batch per rank."""
    from chromoformer_amd import pack
    from tests.synth_data import make_dataset
    import pandas as pd
    meta = make_dataset(str(tmp_path / "npy"), n_genes=37, seed=9)
    out = str(tmp_path / "npy" / pack.DEFAULT_NAME)
    pack.pack(meta, str(tmp_path / "npy"), out, device=None)
    genes = pd.read_csv(meta).gene_id.tolist()
    world, bsz = 2, 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_shard_worker, args=(world, port, out, genes, bsz, ret), nprocs=world, join=True)
    sizes, batches = ret["sizes"], ret["batches"]
    assert sorted(n for n, _ in sizes) == [18, 19] and sum(n for n, _ in sizes) == len(genes)      # 1/world each
    assert len({nb for _, nb in sizes}) == 1 and sizes[0][1] == (37 // 2) // bsz                   # the same batch count everywhere
    seen = set()
    for k in range(sizes[0][1]):
        glob = [g for r in range(world) for g in batches[r][k]]
        assert len(glob) == world * bsz and len(set(glob)) == len(glob) and not (set(glob) & seen)
        seen |= set(glob)
