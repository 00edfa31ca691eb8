"""The CPU oracle against the committed golden vectors (generated from the reference by
tests/golden/make_goldens.py).  This is what pins the oracle; it runs without a GPU."""
import json
import os

import numpy as np
import torch

from oracle import chromoformer_oracle as orc
from oracle import dataset_oracle as dso
from tests.helpers import GOLDEN, checksum, load_npz_batch, take


def test_g3_state_dict_order_shapes_checksums():
    g = json.load(open(os.path.join(GOLDEN, "state_dict.json")))
    assert g["n_params"] == 5342672
    for seed in (42, 123):
        for reg in (False, True):
            ref = g["seed%d_%s" % (seed, "reg" if reg else "clf")]
            P = orc.init_params(None, seed, reg)
            assert list(P.keys()) == ref["keys"]
            assert [list(v.shape) for v in P.values()] == ref["shapes"]
            got = np.array([checksum(v) for v in P.values()])
            np.testing.assert_allclose(got, np.array(ref["checksums"]), rtol=1e-12, atol=1e-12)
    assert len(ref["keys"]) == 370
    assert sum(orc.never_trained(k) for k in ref["keys"]) == 36


def test_g2_known_answers():
    batch, ex = load_npz_batch("kat.npz")
    with torch.no_grad():
        oc = orc.forward(orc.init_params(None, 42, False), batch)
        orr = orc.forward(orc.init_params(None, 42, True), batch)
    assert abs(float(oc.sum()) + 3.1917) < 5e-4          # net.py:558-564
    assert abs(float(orr.sum()) + 0.1900) < 5e-4         # net.py:566-568
    assert np.abs(oc.numpy() - ex["logits_clf"]).max() < 1e-6
    assert np.abs(orr.numpy() - ex["logits_reg"]).max() < 1e-6


def test_g1_g7_demo_subset_logits_and_stages():
    batch, ex = load_npz_batch("demo_subset.npz")
    P = orc.init_params(None, 123, False)
    with torch.no_grad():
        logits, st = orc.forward(P, batch, return_stages=True)
    assert np.abs(logits.numpy() - ex["logits"]).max() < 1e-6
    for b in (2000, 500, 100):
        for k in ("embed_tss", "pairwise", "regulation_row0"):
            got = st["%s.%d" % (k, b)].numpy().reshape(-1)
            assert np.abs(got - ex["g7.%s.%d" % (k, b)].reshape(-1)).max() < 2e-6, (k, b)


def test_g4_train_step():
    sub, _ = load_npz_batch("demo_subset.npz")
    z = np.load(os.path.join(GOLDEN, "train_step.npz"))
    batch = take(sub, list(z["rows_in_demo_subset"]))
    names = list(z["names"])
    for reg in (False, True):
        tag = "reg" if reg else "clf"
        P = orc.init_params(None, 42, reg)
        assert list(P.keys()) == names
        for t in P.values():
            t.requires_grad_(True)
        opt = orc.make_optimizer(P, "3e-5")
        b = dict(batch)
        b["label"] = torch.from_numpy(z[tag + ".label"]).view(-1)
        loss, logits = orc.train_step(P, opt, b, regression=reg)
        assert abs(float(loss) - float(z[tag + ".loss"])) < 1e-6
        assert np.abs(logits.numpy() - z[tag + ".logits"]).max() < 1e-6
        none = z[tag + ".grad_is_none"]
        for i, k in enumerate(names):
            g = P[k].grad
            if none[i]:
                assert orc.never_trained(k) and (g is None or float(g.abs().max()) == 0.0)
                continue
            full = tag + ".grad." + k
            if full in z.files:
                ref = z[full]
                assert np.abs(g.numpy() - ref).max() <= 5e-4 * (np.abs(ref).max() + 1e-12), k
            ref = z[tag + ".grad_checksums"][i]
            got = checksum(g)
            assert abs(got[2] - ref[2]) <= 1e-3 * ref[2] + 1e-18, k   # sum of squares
        after = np.array([checksum(P[k]) for k in names])
        np.testing.assert_allclose(after, z[tag + ".param_checksums_after"], rtol=1e-6, atol=1e-7)
        st = opt.state_dict()["state"]
        assert sorted(st.keys()) == list(z[tag + ".opt_state_keys"])
        assert len(st) == 334


def test_g5_dataset_items():
    z = np.load(os.path.join(GOLDEN, "dataset.npz"))
    import io
    import pandas as pd
    meta = pd.read_csv(io.StringIO(str(z["meta_csv"])))
    raws = {k[4:]: z[k] for k in z.files if k.startswith("raw.")}
    for _, r in meta.iterrows():
        pcs = [] if not isinstance(r["neighbors"], str) else r["neighbors"].split(";")
        pcs = [(p.split(":")[0], int(p.split(":")[1].split("-")[0]), int(p.split(":")[1].split("-")[1])) for p in pcs]
        scs = [] if not isinstance(r["scores"], str) else [float(s) for s in r["scores"].split(";")]
        for w_prom in (40000, 10000):
            it = dso.gene_item(lambda c, s, e: raws["%s:%d-%d" % (c, s, e)], (r["chrom"], r["start"]), r["strand"],
                               pcs, scs, r["label"], w_prom=w_prom)
            tag = "item.clf.w%d.%s" % (w_prom, r["gene_id"])
            for k, v in it.items():
                if isinstance(v, dict):
                    for b, t in v.items():
                        ref = z["%s.%s.%d" % (tag, k, b)]
                        assert tuple(t.shape) == ref.shape
                        if t.dtype == torch.bool:
                            assert np.array_equal(t.numpy(), ref), (tag, k, b)
                        else:
                            assert np.abs(t.numpy() - ref).max() < 2e-6, (tag, k, b)
                else:
                    assert np.allclose(v.numpy(), z["%s.%s" % (tag, k)], atol=1e-6)
        lab = dso.gene_item(lambda c, s, e: raws["%s:%d-%d" % (c, s, e)], (r["chrom"], r["start"]), r["strand"],
                            pcs, scs, np.log2(r["expression"] + 1), regression=True)["label"]
        assert abs(float(lab) - float(z["item.reg.w40000.%s.label" % r["gene_id"]])) < 1e-6
