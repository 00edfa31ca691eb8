"""`python -m chromoformer_amd.train` under torch.distributed.run with TWO ranks (BASELINE configs[2] in miniature): the
whole data-parallel entrypoint -- per-rank store loading, per-rank EpochFeed inside the step graphs, two-bucket gradient
all-reduce, sharded validation + all_gather, rank-0-only checkpoint -- replacing train.py:137-140,163-344.  A one-GPU box
cannot run RCCL with two ranks, so both share device 0 and talk over gloo (CF_SHARE_DEVICE / CF_DIST_BACKEND, the hooks
bench.py has); everything else is the production path.

  * `--dp-shard global`, bsz 4 x 2 ranks must reproduce the single-process bsz-8 run on the G6 synthetic dataset:
    parameters to 2e-6 (sums associate differently), validation outputs in dataset order, one checkpoint + `.done`.
  * `--dp-shard static` completes with every rank holding ~1/world of the training split and the same step count."""
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

from tests.synth_data import make_dataset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(argv, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CF_SHARE_DEVICE="1", CF_DIST_BACKEND="gloo", PYTHONPATH=ROOT)
    for attempt in range(2):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_port()), "-m", "chromoformer_amd.train"] + argv
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
        if r.returncode == 0 or not any(k in r.stderr for k in ("EADDRINUSE", "address already in use", "RendezvousConnectionError", "Connection refused")):
            return r
    return r


def _setup(tmp_path, bsz):
    meta = make_dataset(str(tmp_path / "npy"), n_genes=48, seed=2024)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "chromoformer_amd", "configs", "default.yaml")))
    cfg["bsz"], cfg["num_epoch"] = bsz, 3
    path = str(tmp_path / ("cfg%d.yaml" % bsz))
    yaml.safe_dump(cfg, open(path, "w"))
    return meta, path


def _argv(out, cfg, meta, tmp_path, extra=()):
    return ["-o", out, "-c", cfg, "--exp-id", "dp", "-m", meta, "-d", str(tmp_path / "npy"), "--fold", "0",
            "--binsizes", "2000", "500", "100"] + list(extra)


@pytest.mark.parametrize("reg", [False, True])
def test_two_rank_global_sharding_reproduces_the_single_process_run(tmp_path, reg):
    from chromoformer_amd import train
    meta, cfg8 = _setup(tmp_path, 8)
    _, cfg4 = _setup(tmp_path, 4)
    extra = ["--regression"] if reg else []
    one = str(tmp_path / "one" / "ck.pt")
    os.makedirs(os.path.dirname(one))
    assert train.main(_argv(one, cfg8, meta, tmp_path, extra)) == 0
    two = str(tmp_path / "two" / "ck.pt")
    os.makedirs(os.path.dirname(two))
    open(two + ".done", "w").write("stale marker of an earlier run\n")
    r = _torchrun(_argv(two, cfg4, meta, tmp_path, extra + ["--dp-shard", "global"]))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert sorted(os.listdir(os.path.dirname(two))) == ["ck.pt", "ck.pt.done"]          # rank 0 only, nothing half-written
    assert open(two + ".done").read().startswith("epochs 2")
    a = torch.load(one, map_location="cpu", weights_only=False)
    b = torch.load(two, map_location="cpu", weights_only=False)
    assert list(a.keys()) == list(b.keys()) and a["epoch"] == b["epoch"] == 2
    assert list(a["net"].keys()) == list(b["net"].keys())
    for k in a["net"]:
        assert (a["net"][k] - b["net"][k]).abs().max() <= 2e-6, k
    assert len(b["optimizer"]["state"]) == 334 and b["optimizer"]["param_groups"][0]["lr"] == a["optimizer"]["param_groups"][0]["lr"]
    # validation: 12 genes, 6 per rank in batches of 4 + 2, gathered in dataset order on rank 0
    assert np.array_equal(np.asarray(a["val_label"]), np.asarray(b["val_label"]))
    assert np.abs(np.asarray(a["val_score"]) - np.asarray(b["val_score"])).max() < 1e-5
    assert abs(float(a["last_val_loss"]) - float(b["last_val_loss"])) < 1e-5 * max(1.0, abs(float(a["last_val_loss"])))
    # rank 0 prints the reference's lines once; every rank reports what it holds (the whole split under `global`)
    assert r.stdout.count("Validation loss=") == 2
    held = re.findall(r"\[rank (\d)/2\] dp-shard global: train store (\d+) of (\d+) genes, validation slice (\d+) of (\d+) genes, (\d+) steps", r.stderr)
    assert sorted(held) == [("0", "36", "36", "6", "12", "4"), ("1", "36", "36", "6", "12", "4")]


def test_two_rank_static_sharding_completes_with_one_half_of_the_split_per_rank(tmp_path):
    meta, cfg4 = _setup(tmp_path, 4)
    out = str(tmp_path / "ck.pt")
    r = _torchrun(_argv(out, cfg4, meta, tmp_path, ["--dp-shard", "static"]))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    held = re.findall(r"\[rank (\d)/2\] dp-shard static: train store (\d+) of (\d+) genes, validation slice (\d+) of (\d+) genes, (\d+) steps", r.stderr)
    assert sorted(held) == [("0", "18", "36", "6", "12", "4"), ("1", "18", "36", "6", "12", "4")]
    c = torch.load(out, map_location="cpu", weights_only=False)
    assert c["epoch"] == 2 and os.path.exists(out + ".done") and not os.path.exists(out + ".tmp")
    assert len(c["optimizer"]["state"]) == 334 and float(next(iter(c["optimizer"]["state"].values()))["step"]) == 8.0
    assert np.asarray(c["val_score"]).shape == (12,) and np.isfinite(np.asarray(c["val_score"])).all()
    # the parameters moved and are finite
    from chromoformer_amd import ChromoformerClassifier
    ref = ChromoformerClassifier(seed=42).state_dict()
    moved = sum(float((c["net"][k] - ref[k].cpu()).abs().max()) > 0 for k in c["net"])
    assert moved >= 334 and all(bool(torch.isfinite(v).all()) for v in c["net"].values())
