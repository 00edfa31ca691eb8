"""ChromoformerDataset / GeneStore against golden G5 (full __getitem__ outputs of the reference on
synthetic raw regions: '-' strand gene, partial last bin, 40-kb pCRE, gene without partners)."""
import io
import os

import numpy as np
import pandas as pd
import pytest
import torch

from chromoformer_amd.data import ChromoformerDataset, GeneStore, shard_indices
from tests.helpers import GOLDEN


@pytest.fixture(scope="module")
def synth(tmp_path_factory):
    z = np.load(os.path.join(GOLDEN, "dataset.npz"))
    d = tmp_path_factory.mktemp("npy")
    for k in z.files:
        if k.startswith("raw."):
            np.save(os.path.join(d, k[4:] + ".npy"), z[k])
    meta = os.path.join(d, "meta.csv")
    open(meta, "w").write(str(z["meta_csv"]))
    return z, str(d), meta


def test_getitem_matches_reference_items(synth):
    z, npy_dir, meta = synth
    genes = pd.read_csv(meta).gene_id.tolist()
    for w_prom in (40000, 10000):
        ds = ChromoformerDataset(meta, npy_dir, genes, w_prom=w_prom)
        assert len(ds) == 3
        for i, gene in enumerate(genes):
            it = ds[i]
            tag = "item.clf.w%d.%s" % (w_prom, gene)
            for k, v in it.items():
                if isinstance(v, dict):
                    for b, t in v.items():
                        ref = z["%s.%s.%d" % (tag, k, b)]
                        assert tuple(t.shape) == ref.shape and str(t.dtype).endswith(str(ref.dtype)), (k, b)
                        if t.dtype == torch.bool:
                            assert np.array_equal(t.numpy(), ref), (tag, k, b)
                        else:
                            assert np.abs(t.numpy() - ref).max() < 2e-6, (tag, k, b)
                else:
                    assert np.allclose(v.numpy(), z["%s.%s" % (tag, k)], atol=1e-6)
    reg = ChromoformerDataset(meta, npy_dir, genes, regression=True)
    for i, gene in enumerate(genes):
        assert abs(float(reg[i]["label"]) - float(z["item.reg.w40000.%s.label" % gene])) < 1e-6
        assert reg[i]["label"].dtype == torch.float32


def test_store_is_the_compact_form_of_getitem(synth):
    z, npy_dir, meta = synth
    genes = pd.read_csv(meta).gene_id.tolist()
    ds = ChromoformerDataset(meta, npy_dir, genes, binsizes=["2000", "500", "100"])      # CLI strings accepted
    store = GeneStore(ds, pin=False)
    b = store.batch([2, 0, 1])
    for slot, i in enumerate([2, 0, 1]):
        it = ds[i]
        for bs in (2000, 500, 100):
            L = 40000 // bs
            assert torch.equal(b["promoter_feats"][bs][slot], it["promoter_feats"][bs])
            assert torch.equal(b["pcre_feats"][bs][slot], it["pcre_feats"][bs])
            assert torch.equal(b["promoter_pad_masks"][bs][slot].bool(), it["promoter_pad_masks"][bs][0, 0, L // 2])
            assert torch.equal(b["pcre_pad_masks"][bs][slot].bool(), it["pcre_pad_masks"][bs][:, 0, L // 2])
            assert torch.equal(b["interaction_masks"][bs][slot].bool(), it["interaction_masks"][bs][0])
        assert torch.equal(b["interaction_freq"][slot], it["interaction_freq"])
        assert int(b["label"][slot]) == int(it["label"])


def test_oversized_pcre_is_an_error(synth):
    _, npy_dir, meta = synth
    genes = pd.read_csv(meta).gene_id.tolist()
    ds = ChromoformerDataset(meta, npy_dir, genes, w_max=20000)
    with pytest.raises(ValueError):
        ds[1]


def test_shard_indices_partition_global_batches():
    perm = list(range(103))
    for world in (1, 2, 4):
        shards = [shard_indices(perm, r, world, 8) for r in range(world)]
        n_steps = 103 // (8 * world)
        assert all(len(s) == n_steps for s in shards)
        for k in range(n_steps):
            got = sum((list(shards[r][k]) for r in range(world)), [])
            assert got == perm[k * 8 * world:(k + 1) * 8 * world]
    tail = [shard_indices(perm, r, 2, 8, drop_last=False) for r in range(2)]
    assert sorted(sum((list(b) for s in tail for b in s), [])) == perm
