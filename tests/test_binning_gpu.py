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


def test_multi_job_record_matches_the_c_struct():
    from chromoformer_amd.data import BIN_JOB_MULTI
    assert BIN_JOB_MULTI.itemsize == 80 and BIN_JOB_MULTI.fields["out"][1] == 32 and BIN_JOB_MULTI.fields["mask"][1] == 56


@pytest.mark.gpu
@pytest.mark.parametrize("binsizes", [(2000, 500, 100), (500, 100), (2000, 300, 100), (1000, 200, 40)])
def test_one_pass_kernel_equals_the_per_resolution_kernel(binsizes):
    """cf_bin_regions_multi (all resolutions of a region in one launch; nested bin sizes in one pass over the raw bytes) against
    one cf_bin_regions launch per resolution on the same raw regions: rows of every alignment class (multiples of 4 samples ->
    the one-pass path, odd lengths -> the per-resolution walk inside the same launch), windows that start inside the file,
    partial last bins at every resolution, one-sample and empty-pad cases, mirrored regions; (2000, 300, 100) does not nest and
    takes the per-resolution walk for every region.  Mask bytes equal, features within 1e-6 (fp32 additions in another order)."""
    import ctypes as C
    from chromoformer_amd import _lib
    from chromoformer_amd.data import BIN_JOB, BIN_JOB_MULTI
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    F, W = 7, 40000
    lens = [40000, 40000, 4, 100, 104, 1996, 2000, 2004, 12344, 12345, 1833, 2001, 39996, 20000, 500, 96, 7, 3999, 8000, 36]
    lens += [int(4 * rng.integers(1, 10000)) for _ in range(20)]
    raws, jobs = [], []
    for k, ln in enumerate(lens):
        a = (rng.random((F, ln)) * rng.choice([0.5, 4.0, 60.0])).astype(np.float16)
        a[:, rng.random(ln) < 0.3] = 0
        raws.append(a)
        col0 = int(rng.choice([0, 0, 4, 15000])) if ln == 40000 else 0
        ncols = (10000 if col0 == 15000 else ln - col0)
        jobs.append((ln, col0, ncols, k % 2))
    flat = torch.from_numpy(np.concatenate([a.reshape(-1) for a in raws])).to(dev)
    lib = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    nres = len(binsizes)
    Ls = [W // b for b in binsizes]
    out_m = [torch.full((len(lens), L, F), 7.0, device=dev) for L in Ls]
    msk_m = [torch.full((len(lens), L), 9, dtype=torch.uint8, device=dev) for L in Ls]
    out_s = [torch.full((len(lens), L, F), -7.0, device=dev) for L in Ls]
    msk_s = [torch.full((len(lens), L), 5, dtype=torch.uint8, device=dev) for L in Ls]
    mj = np.zeros(len(lens), dtype=BIN_JOB_MULTI)
    off = 0
    for k, (ln, col0, ncols, flip) in enumerate(jobs):
        mj[k]["raw"], mj[k]["ld"], mj[k]["col0"], mj[k]["ncols"], mj[k]["flip"] = flat.data_ptr() + 2 * off, ln, col0, ncols, flip
        for r in range(nres):
            mj[k]["out"][r], mj[k]["mask"][r] = out_m[r][k].data_ptr(), msk_m[r][k].data_ptr()
        off += F * ln
    tab = torch.from_numpy(mj.view(np.uint8)).to(dev)
    bs, nb = (C.c_int * nres)(*binsizes), (C.c_int * nres)(*Ls)
    _lib.check(lib.cf_bin_regions_multi(C.c_void_p(tab.data_ptr()), len(lens), F, nres, bs, nb, max(j[2] for j in jobs), st), "cf_bin_regions_multi")
    for r, b in enumerate(binsizes):
        sj = np.zeros(len(lens), dtype=BIN_JOB)
        for k in range(len(lens)):
            for name in ("raw", "ld", "col0", "ncols", "flip"):
                sj[k][name] = mj[k][name]
            sj[k]["out"], sj[k]["mask"] = out_s[r][k].data_ptr(), msk_s[r][k].data_ptr()
        t1 = torch.from_numpy(sj.view(np.uint8)).to(dev)
        _lib.check(lib.cf_bin_regions(C.c_void_p(t1.data_ptr()), len(lens), F, b, Ls[r], st), "cf_bin_regions")
    torch.cuda.synchronize()
    for r, b in enumerate(binsizes):
        assert torch.equal(msk_m[r], msk_s[r]), (b, "mask")
        d = (out_m[r] - out_s[r]).abs().max(dim=2).values.max(dim=1).values
        assert float(d.max()) < 1e-6, (b, int(d.argmax()), lens[int(d.argmax())], float(d.max()))
    # and against the definition on one region (data.py:68-100), fp64 on the host
    k = 8                                                    # 12,344 samples: partial last bin at every resolution
    x = raws[k].astype(np.float64)
    for r, b in enumerate(binsizes):
        n = -(-x.shape[1] // b)
        ref = np.stack([np.log(x[:, i * b:(i + 1) * b].mean(axis=1) + 1) for i in range(n)], axis=0)      # [n, F]
        left = -(-(Ls[r] - n) // 2)
        got = out_m[r][k].cpu().numpy()
        assert np.abs(got[left:left + n] - ref).max() < 2e-6 and np.all(got[:left] == 0) and np.all(got[left + n:] == 0)
