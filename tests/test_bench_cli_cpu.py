"""bench.py's command-line contract where no GPU is needed: `--gpus N` without a torch.distributed.run environment must not ask the
caller to wrap it in a launcher -- it starts its own ranks, and where fewer than N GPUs are visible it says so and exits non-zero
BEFORE touching a device (the driver's command shape is `python3 bench.py --gpus N --steps K --warmup W`)."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_a_launcher_names_the_visible_devices():
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("a multi-GPU box: the command would run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "CF_SHARE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env, capture_output=True, text=True,
                       timeout=300)
    out = r.stdout + r.stderr
    assert r.returncode != 0
    assert "GPU(s) visible" in out and "launch with" not in out
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]          # no record for a run that did not happen


def test_world_size_mismatch_is_refused_by_name():
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stdout + r.stderr)
