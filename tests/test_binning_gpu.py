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
