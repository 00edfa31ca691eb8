"""cf_bin_regions (HIP replacement of ChromoformerDataset._bin_and_pad + strand flip, data.py:68-113) against the
host binning, which golden G5 pins to the reference's own __getitem__: whole stores on a synthetic dataset with
both strands, genes without partners, partial last bins and a narrowed promoter window; mask bytes bit-exact,
features within 2e-6 (fp32 sums in a different order, then log(1 + x))."""
import numpy as np
import pandas as pd
import pytest
import torch

from tests.synth_data import make_dataset


def test_job_record_matches_the_c_struct():
    from chromoformer_amd.data import BIN_JOB
    assert BIN_JOB.itemsize == 48 and BIN_JOB.fields["out"][1] == 32 and BIN_JOB.fields["mask"][1] == 40


@pytest.mark.gpu
@pytest.mark.parametrize("w_prom", [40000, 10000])
def test_device_binned_store_equals_host_binned_store(tmp_path, w_prom):
    from chromoformer_amd.data import ChromoformerDataset, GeneStore
    meta = make_dataset(str(tmp_path / "npy"), n_genes=24, seed=5)
    genes = pd.read_csv(meta).gene_id.tolist()
    assert set(pd.read_csv(meta).strand) == {"+", "-"}
    ds = ChromoformerDataset(meta, str(tmp_path / "npy"), genes, w_prom=w_prom)
    host = GeneStore(ds, pin=False)
    for kw in (dict(resident=False), dict(resident=True), dict(resident=True, chunk_bytes=1 << 20)):     # several flushes
        dev = GeneStore(ds, device="cuda:0", **kw)
        for r in range(3):
            for name in ("pm", "cm"):
                assert torch.equal(getattr(dev, name)[r].cpu(), getattr(host, name)[r]), (name, r)
            for name in ("pf", "cf"):
                a, b = getattr(dev, name)[r].cpu(), getattr(host, name)[r]
                assert a.shape == b.shape and (a - b).abs().max() < 2e-6, (name, r, float((a - b).abs().max()))
        assert torch.equal(dev.im.cpu(), host.im) and torch.equal(dev.freq.cpu(), host.freq) and torch.equal(dev.label.cpu(), host.label)
        bd, bh = dev.batch([3, 0, 7]), host.batch([3, 0, 7])
        assert bd["pcre_feats"][500].is_cuda == bool(kw["resident"])
        assert (bd["pcre_feats"][500].cpu() - bh["pcre_feats"][500]).abs().max() < 2e-6


@pytest.mark.gpu
def test_oversized_region_is_an_error_on_the_device_path_too(tmp_path):
    from chromoformer_amd.data import ChromoformerDataset, GeneStore
    meta = make_dataset(str(tmp_path / "npy"), n_genes=6, seed=5)
    genes = pd.read_csv(meta).gene_id.tolist()
    ds = ChromoformerDataset(meta, str(tmp_path / "npy"), genes, w_prom=4000, w_max=4000)      # promoters fit, most pCREs (median 5.9 kb) do not
    with pytest.raises(ValueError):
        GeneStore(ds, device="cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("w_prom", [40000, 10000])
def test_kernel_against_the_reference_items_of_golden_g5(tmp_path, w_prom):
    """cf_bin_regions fed with the raw regions of tests/golden/dataset.npz against the tensors the REFERENCE's
    ChromoformerDataset.__getitem__ produced from them (data.py:68-113, 124-212): '+' and '-' strand promoters, the narrowed
    window, partial last bins (1,833 / 2,001 / 12,345 samples), a 100-sample and a 40-kb pCRE, a gene without partners."""
    import os
    from chromoformer_amd.data import ChromoformerDataset, GeneStore
    from tests.helpers import GOLDEN
    z = np.load(os.path.join(GOLDEN, "dataset.npz"))
    d = tmp_path / "npy"
    d.mkdir()
    for k in z.files:
        if k.startswith("raw."):
            np.save(str(d / (k[4:] + ".npy")), z[k])
    meta = str(d / "meta.csv")
    open(meta, "w").write(str(z["meta_csv"]))
    genes = pd.read_csv(meta).gene_id.tolist()
    ds = ChromoformerDataset(meta, str(d), genes, w_prom=w_prom)
    store = GeneStore(ds, device="cuda:0", resident=True)
    for i, gene in enumerate(genes):
        tag = "item.clf.w%d.%s" % (w_prom, gene)
        for r, b in enumerate((2000, 500, 100)):
            L = 40000 // b
            assert np.abs(store.pf[r][i].cpu().numpy() - z["%s.promoter_feats.%d" % (tag, b)]).max() < 2e-6, (gene, b)
            assert np.abs(store.cf[r][i].cpu().numpy() - z["%s.pcre_feats.%d" % (tag, b)]).max() < 2e-6, (gene, b)
            assert np.array_equal(store.pm[r][i].cpu().numpy().astype(bool), z["%s.promoter_pad_masks.%d" % (tag, b)][0, 0, L // 2]), (gene, b)
            assert np.array_equal(store.cm[r][i].cpu().numpy().astype(bool), z["%s.pcre_pad_masks.%d" % (tag, b)][:, 0, L // 2]), (gene, b)
            assert np.array_equal(store.im[i].cpu().numpy().astype(bool), z["%s.interaction_masks.%d" % (tag, b)][0]), (gene, b)
        assert np.allclose(store.freq[i].cpu().numpy(), z["%s.interaction_freq" % tag], atol=1e-6)
        assert int(store.label[i]) == int(z["%s.label" % tag])
