"""The CV-sweep scheduler running real trainings on the GPU (chromoformer/Snakefile:25, 48-62): two cell lines x one fold,
one `chromoformer_amd.train` process per job pinned with HIP_VISIBLE_DEVICES, checkpoints under the Snakefile's naming
scheme, equal to what a direct `train.main` call on the same inputs writes."""
import os
import shutil

import numpy as np
import pytest
import torch
import yaml

from tests.synth_data import make_dataset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("regression,per_gpu", [(False, 1), (True, 1), (False, 2)], ids=["classifier", "regressor", "classifier_two_jobs_at_once"])
def test_sweep_runs_jobs_and_names_checkpoints_like_the_snakefile(tmp_path, regression, per_gpu):
    """(regressor: BASELINE.json configs[4], `--regression` through the sweep -- Chromoformer-reg per cell line and fold;
    two_jobs_at_once: `--jobs-per-gpu 2`, both trainings share the device while they run -- each still equals its direct run bit for bit)"""
    from chromoformer_amd import sweep, train
    extra = ["--regression"] if regression else []
    cfg = yaml.safe_load(open(os.path.join(ROOT, "chromoformer_amd", "configs", "default.yaml")))
    cfg["bsz"], cfg["num_epoch"] = 8, 2
    cfg_path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    for eid, seed in (("E003", 11), ("E004", 12)):
        os.makedirs(tmp_path / "data" / eid)
        meta = make_dataset(str(tmp_path / "data" / eid / "npy"), n_genes=40, seed=seed)
        shutil.move(meta, str(tmp_path / "data" / eid / "train.csv"))
    out_dir = str(tmp_path / "ckpts")
    argv = ["--meta-template", str(tmp_path / "data" / "{eid}" / "train.csv"), "--npy-dir-template", str(tmp_path / "data" / "{eid}" / "npy"),
            "-c", cfg_path, "--exp-id", "exp", "--conf", "1", "--eids", "E003", "E004", "--folds", "1", "--gpus", "1",
            "--out-dir", out_dir, "--poll", "0.2", "--jobs-per-gpu", str(per_gpu)] + extra
    env_before = os.environ.get("PYTHONPATH")
    os.environ["PYTHONPATH"] = ROOT + (os.pathsep + env_before if env_before else "")
    try:
        assert sweep.main(argv) == 0
    finally:
        if env_before is None:
            os.environ.pop("PYTHONPATH")
        else:
            os.environ["PYTHONPATH"] = env_before
    for eid in ("E003", "E004"):
        ck = os.path.join(out_dir, eid, "exp-%s-conf1-fold1.pt" % eid)          # Snakefile:25
        assert os.path.exists(ck) and os.path.exists(ck + ".done"), open(ck + ".log").read()[-2000:]
        direct = str(tmp_path / ("direct-%s.pt" % eid))
        assert train.main(["-o", direct, "-c", cfg_path, "--exp-id", "exp", "-m", str(tmp_path / "data" / eid / "train.csv"),
                           "-d", str(tmp_path / "data" / eid / "npy"), "--fold", "1"] + extra) == 0
        a, b = torch.load(ck, map_location="cpu", weights_only=False), torch.load(direct, map_location="cpu", weights_only=False)
        assert a["epoch"] == b["epoch"] == 1 and list(a["net"]) == list(b["net"])
        for k in a["net"]:
            assert torch.equal(a["net"][k], b["net"][k]), k          # deterministic kernels: the replica equals the direct run bit for bit
        assert np.array_equal(a["val_score"], b["val_score"])
    # a second sweep finds both jobs finished and launches nothing
    launched = []
    args = type("A", (), dict(config=cfg_path, exp_id="exp", meta_template="", npy_dir_template="", binsizes=None, regression=regression, gpus=1, poll=0.0))
    done = sweep.run(sweep.plan(["E003", "E004"], ["1"], "exp", "1", out_dir), args, launch=lambda *a, **k: launched.append(a))
    assert not launched and set(done.values()) == {0}
