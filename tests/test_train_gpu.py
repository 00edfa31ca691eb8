"""End-to-end drop-in check of `python -m chromoformer_amd.train` against golden G6: the
reference's own train.py run (CPU, in the build container) on the deterministic synthetic dataset
of tests/synth_data.py -- same shuffling, same 2 epochs x 4 steps, same checkpoint."""
import os

import numpy as np
import pytest
import torch
import yaml

from tests.helpers import GOLDEN, checksum
from tests.synth_data import make_dataset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("reg", [False, True])
def test_training_run_matches_reference_checkpoint(tmp_path, reg):
    from chromoformer_amd import train
    z = np.load(os.path.join(GOLDEN, "train_run.npz"))
    tag = "reg" if reg else "clf"
    meta = make_dataset(str(tmp_path / "npy"), n_genes=48, seed=2024)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "chromoformer_amd", "configs", "default.yaml")))
    cfg["bsz"], cfg["num_epoch"] = 8, 3
    cfg_path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    out = str(tmp_path / "ck.pt")
    argv = ["-o", out, "-c", cfg_path, "--exp-id", "g6", "-m", meta, "-d", str(tmp_path / "npy"), "--fold", "0",
            "--binsizes", "2000", "500", "100"]
    if reg:
        argv.append("--regression")
    assert train.main(argv) == 0
    c = torch.load(out, map_location="cpu", weights_only=False)
    assert list(c.keys()) == list(z[tag + ".ckpt_keys"])
    assert c["epoch"] == int(z[tag + ".epoch"])
    assert abs(c["optimizer"]["param_groups"][0]["lr"] - float(z[tag + ".lr"])) < 1e-12
    assert len(c["optimizer"]["state"]) == int(z[tag + ".opt_n_state"]) == 334
    # the reference saves an AdamW that has a StepLR attached (train.py:157-158): same key set, initial_lr included
    ref_opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(1))], lr=3e-5)
    torch.optim.lr_scheduler.StepLR(ref_opt, step_size=1, gamma=0.87)
    assert set(c["optimizer"]["param_groups"][0]) == set(ref_opt.state_dict()["param_groups"][0])
    assert c["optimizer"]["param_groups"][0]["initial_lr"] == 3e-5
    assert os.path.exists(out + ".done") and not os.path.exists(out + ".tmp")
    st = next(iter(c["optimizer"]["state"].values()))
    assert float(st["step"]) == float(z[tag + ".opt_step"]) and st["step"].dtype == torch.float32 and st["step"].dim() == 0
    assert str(np.asarray(c["val_score"]).dtype) == str(z[tag + ".val_score_dtype"])
    assert str(np.asarray(c["val_label"]).dtype) == str(z[tag + ".val_label_dtype"])
    assert np.array_equal(np.asarray(c["val_label"]), z[tag + ".val_label"])
    # 8 AdamW steps from identical weights, identical batches: the validation outputs agree closely
    assert np.abs(np.asarray(c["val_score"]) - z[tag + ".val_score"]).max() < (5e-3 if reg else 1e-3)
    assert abs(float(c["last_val_loss"]) - float(z[tag + ".last_val_loss"])) < 2e-3 * max(1.0, float(z[tag + ".last_val_loss"]))
    got = np.array([checksum(v) for v in c["net"].values()])
    ref = z[tag + ".param_checksums"]
    assert np.abs(got[:, 2] - ref[:, 2]).max() <= 1e-4 * ref[:, 2].max()
    # the checkpoint loads back into both model classes' load_state_dict
    from chromoformer_amd import ChromoformerClassifier, ChromoformerRegressor
    m = (ChromoformerRegressor if reg else ChromoformerClassifier)()
    m.load_state_dict(c["net"])


@pytest.mark.parametrize("d_emb", [128, 256, 64])
def test_an_edited_config_trains(tmp_path, d_emb):
    """configs/default.yaml:14-31 with other head counts / Regulation width / row width (the shapes of tests/test_config_variants_gpu.py, checked there against
    the oracle) through the training entrypoint: constructs, trains, validates, saves a checkpoint of the edited shapes that loads back.  `d_emb` is the
    Embedding's d_model in the entrypoint (train.py:266 here, train.py of the reference likewise); the Pairwise d_model has to follow it (net.py:361-370)."""
    from chromoformer_amd import train, ChromoformerClassifier
    meta = make_dataset(str(tmp_path / "npy"), n_genes=48, seed=2024)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "chromoformer_amd", "configs", "default.yaml")))
    cfg["bsz"], cfg["num_epoch"] = 8, 2
    cfg["embed"]["n_heads"] = 4 if d_emb == 128 else 2
    cfg["pairwise_interaction"]["n_heads"] = 1
    cfg["embed"]["d_model"] = cfg["pairwise_interaction"]["d_model"] = d_emb
    cfg["regulation"].update(n_heads=4, d_model=128, n_layers=3)
    cfg_path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    out = str(tmp_path / "ck.pt")
    assert train.main(["-o", out, "-c", cfg_path, "--exp-id", "edited", "-m", meta, "-d", str(tmp_path / "npy"), "--fold", "0",
                       "--binsizes", "2000", "500", "100"]) == 0
    c = torch.load(out, map_location="cpu", weights_only=False)
    net = c["net"]
    assert net["regulation.2000.transformer.layers.0.self_att.att.weight"].shape == (512, d_emb)      # (4 chunks x d_model 128 rows; the input is d_emb wide)
    assert net["regulation.2000.transformer.layers.0.self_att.gamma_f"].shape == (4,)
    assert net["embed.100.transformer.layers.0.self_att.gamma_f"].shape == (4 if d_emb == 128 else 2,)
    assert net["embed.100.transformer.layers.0.self_att.att.weight"].shape[1] == d_emb and net["embed.100.lin_proj.weight"].shape[0] == d_emb
    assert "regulation.2000.transformer.layers.3.ff.l1.weight" not in net
    assert np.isfinite(float(c["last_val_loss"])) and all(torch.isfinite(v).all() for v in net.values())
    m = ChromoformerClassifier(cfg["n_feats"] if "n_feats" in cfg else 7, cfg["embed"]["d_model"], cfg.get("d_head", 128), cfg["embed"], cfg["pairwise_interaction"],
                               cfg["regulation"], binsizes=[2000, 500, 100], i_max=cfg.get("i_max", 8), w_max=cfg.get("w_max", 40000))
    m.load_state_dict(net)
