"""The packed store (chromoformer_amd.pack): pack on the host, reopen as memory maps, gather splits -- equal to binning the
same genes directly, and pinned to the reference's items of golden G5."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from chromoformer_amd import pack
from chromoformer_amd.data import ChromoformerDataset, GeneStore
from tests.helpers import GOLDEN
from tests.synth_data import make_dataset


def _same(a, b):
    for r in range(3):
        for n in ("pf", "cf", "pm", "cm"):
            assert torch.equal(getattr(a, n)[r].cpu(), getattr(b, n)[r].cpu()), (n, r)
    assert torch.equal(a.im.cpu(), b.im.cpu()) and torch.equal(a.freq.cpu(), b.freq.cpu()) and torch.equal(a.label.cpu(), b.label.cpu())


def test_pack_roundtrip_and_split_gather(tmp_path):
    meta = make_dataset(str(tmp_path / "npy"), n_genes=20, seed=3)
    out = str(tmp_path / "npy" / pack.DEFAULT_NAME)
    assert pack.main(["-m", meta, "-d", str(tmp_path / "npy"), "-o", out, "--host"]) == 0
    ps = pack.PackedStore(out)
    genes = pd.read_csv(meta).gene_id.tolist()
    assert ps.genes == genes and ps.matches([2000, 500, 100], 8, 40000, 40000, 7) and not ps.matches([2000, 500, 100], 8, 10000, 40000, 7)
    split = [genes[i] for i in (7, 2, 19, 0, 11)]                 # a split is any subset in any order
    ds = ChromoformerDataset(meta, str(tmp_path / "npy"), split)
    _same(ps.store(split), GeneStore(ds, pin=False))
    reg = ps.store(split, regression=True)
    want = ChromoformerDataset(meta, str(tmp_path / "npy"), split, regression=True)
    assert torch.allclose(reg.label, torch.tensor([float(want[i]["label"]) for i in range(len(split))]), atol=1e-6) and reg.label.dtype == torch.float32
    assert len(ps.store([])) == 0
    with pytest.raises(KeyError):
        ps.store(["NOT_A_GENE"])
    # lookup rules: default name next to the signals, signature must match
    assert pack.find(str(tmp_path / "npy"), None, [2000, 500, 100], 8, 40000, 40000, 7, genes) is not None
    assert pack.find(str(tmp_path / "npy"), None, [2000, 500, 100], 8, 10000, 40000, 7, genes) is None
    with pytest.raises(ValueError):
        pack.find(str(tmp_path / "npy"), out, [1000, 500, 100], 8, 40000, 40000, 7, genes)


def test_packed_items_equal_the_reference_items_of_golden_g5(tmp_path):
    z = np.load(os.path.join(GOLDEN, "dataset.npz"))
    d = tmp_path / "npy"
    d.mkdir()
    for k in z.files:
        if k.startswith("raw."):
            np.save(str(d / (k[4:] + ".npy")), z[k])
    meta = str(d / "meta.csv")
    open(meta, "w").write(str(z["meta_csv"]))
    out = str(d / pack.DEFAULT_NAME)
    pack.pack(meta, str(d), out, device=None)
    genes = pd.read_csv(meta).gene_id.tolist()
    store = pack.PackedStore(out).store(genes[::-1])
    for slot, gene in enumerate(genes[::-1]):
        tag = "item.clf.w40000.%s" % gene
        for r, b in enumerate((2000, 500, 100)):
            L = 40000 // b
            assert np.abs(store.pf[r][slot].numpy() - z["%s.promoter_feats.%d" % (tag, b)]).max() < 2e-6
            assert np.abs(store.cf[r][slot].numpy() - z["%s.pcre_feats.%d" % (tag, b)]).max() < 2e-6
            assert np.array_equal(store.cm[r][slot].numpy().astype(bool), z["%s.pcre_pad_masks.%d" % (tag, b)][:, 0, L // 2])
            assert np.array_equal(store.pm[r][slot].numpy().astype(bool), z["%s.promoter_pad_masks.%d" % (tag, b)][0, 0, L // 2])
        assert int(store.label[slot]) == int(z["%s.label" % tag])
