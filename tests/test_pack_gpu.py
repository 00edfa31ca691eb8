"""The packed store on the GPU: packed with cf_bin_regions, loaded straight into HBM, pinned to the reference's items of
golden G5; training through it equals training from the raw .npy files bit for bit."""
import os

import numpy as np
import pandas as pd
import pytest
import torch
import yaml

from tests.helpers import GOLDEN
from tests.synth_data import make_dataset

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpu_packed_store_equals_reference_items(tmp_path):
    from chromoformer_amd import pack
    z = np.load(os.path.join(GOLDEN, "dataset.npz"))
    d = tmp_path / "npy"
    d.mkdir()
    for k in z.files:
        if k.startswith("raw."):
            np.save(str(d / (k[4:] + ".npy")), z[k])
    meta = str(d / "meta.csv")
    open(meta, "w").write(str(z["meta_csv"]))
    for w_prom in (40000, 10000):
        out = str(d / ("w%d.cfstore" % w_prom))
        assert pack.main(["-m", meta, "-d", str(d), "-o", out, "--w-prom", str(w_prom)]) == 0          # bins on the GPU
        genes = pd.read_csv(meta).gene_id.tolist()
        store = pack.PackedStore(out).store(genes[::-1], device=torch.device("cuda", 0))
        assert store.pf[2].is_cuda and store.struct().n_genes == 3
        for slot, gene in enumerate(genes[::-1]):
            tag = "item.clf.w%d.%s" % (w_prom, gene)
            for r, b in enumerate((2000, 500, 100)):
                L = 40000 // b
                assert np.abs(store.pf[r][slot].cpu().numpy() - z["%s.promoter_feats.%d" % (tag, b)]).max() < 2e-6
                assert np.abs(store.cf[r][slot].cpu().numpy() - z["%s.pcre_feats.%d" % (tag, b)]).max() < 2e-6
                assert np.array_equal(store.pm[r][slot].cpu().numpy().astype(bool), z["%s.promoter_pad_masks.%d" % (tag, b)][0, 0, L // 2])
                assert np.array_equal(store.cm[r][slot].cpu().numpy().astype(bool), z["%s.pcre_pad_masks.%d" % (tag, b)][:, 0, L // 2])
                assert np.array_equal(store.im[slot].cpu().numpy().astype(bool), z["%s.interaction_masks.%d" % (tag, b)][0])
            assert np.allclose(store.freq[slot].cpu().numpy(), z["%s.interaction_freq" % tag], atol=1e-6)


def test_training_from_the_packed_store_equals_training_from_npy_files(tmp_path):
    from chromoformer_amd import pack, train
    meta = make_dataset(str(tmp_path / "npy"), n_genes=48, seed=2024)
    cfg = yaml.safe_load(open(os.path.join(ROOT, "chromoformer_amd", "configs", "default.yaml")))
    cfg["bsz"], cfg["num_epoch"] = 8, 2
    cfg_path = str(tmp_path / "cfg.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    base = ["-c", cfg_path, "--exp-id", "p", "-m", meta, "-d", str(tmp_path / "npy"), "--fold", "0"]
    assert train.main(["-o", str(tmp_path / "a.pt")] + base) == 0                      # no store yet: raw files, GPU binning
    assert pack.main(["-m", meta, "-d", str(tmp_path / "npy")]) == 0                   # default name next to the signals
    assert train.main(["-o", str(tmp_path / "b.pt")] + base) == 0                      # picked up automatically
    a, b = (torch.load(str(tmp_path / n), map_location="cpu", weights_only=False) for n in ("a.pt", "b.pt"))
    for k in a["net"]:
        assert torch.equal(a["net"][k], b["net"][k]), k
    assert np.array_equal(a["val_score"], b["val_score"]) and np.array_equal(a["val_label"], b["val_label"])
