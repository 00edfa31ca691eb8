"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (needs /root/reference); the fixtures it
writes are plain data (inputs + expected outputs) and travel to the GPU box, the
reference itself never does.  While generating, every golden is also checked
against the CPU oracle (oracle/), which is how the oracle is pinned.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens.py

Goldens:
  G1 demo_subset.npz     untrained seed-123 demo forward (all 100 genes checked against
                         demo/random_prediction.out here; 6-gene subset committed)
  G2 kat.npz             known-answer smoke block of net.py (-3.1917 / -0.1900)
  G3 state_dict.json     key / shape / order + per-tensor checksums, seeds 42 & 123
  G4 train_step.npz      one optimisation step on a fixed 4-gene batch
  G5 dataset.npz         __getitem__ on synthetic raw regions (incl. '-' strand, 0 partners)
  G7 (inside G1)         per-stage intermediates
  G8 (inside G1)         structural-exactness witnesses (dead rows / dummy pCREs)
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import numpy as np
import pandas as pd
import torch

torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self
sys.modules.setdefault("wandb", types.ModuleType("wandb"))

from chromoformer.net import Chromoformer, ChromoformerClassifier, ChromoformerRegressor  # noqa: E402
from chromoformer.data import ChromoformerDataset  # noqa: E402

from oracle import chromoformer_oracle as orc  # noqa: E402
from oracle import dataset_oracle as dso  # noqa: E402

BINS = [2000, 500, 100]
torch.set_num_threads(8)


def kws():
    return ({"n_layers": 1, "n_heads": 2, "d_model": 128, "d_ff": 128},
            {"n_layers": 2, "n_heads": 2, "d_model": 128, "d_ff": 256},
            {"n_layers": 6, "n_heads": 8, "d_model": 256, "d_ff": 256})


def collate(items):
    out = {}
    for k in items[0]:
        if isinstance(items[0][k], dict):
            out[k] = {b: torch.stack([it[k][b] for it in items]) for b in items[0][k]}
        else:
            out[k] = torch.stack([it[k] for it in items])
    return out


def flat_save(path, batch, extra):
    arrs = {}
    for k, v in batch.items():
        if isinstance(v, dict):
            for b, t in v.items():
                arrs["%s.%d" % (k, b)] = t.numpy()
        else:
            arrs[k] = v.numpy()
    arrs.update(extra)
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024))


def ref_call(model, batch):
    return model(batch["promoter_feats"], batch["promoter_pad_masks"], batch["pcre_feats"],
                 batch["pcre_pad_masks"], batch["interaction_masks"], batch["interaction_freq"])


def sd_as_params(model):
    return {k: v.detach().clone() for k, v in model.state_dict().items()}


def checksum(t):
    t = t.detach().double()
    return [float(t.sum()), float(t.abs().sum()), float((t * t).sum())]


# --------------------------------------------------------------------------- #
def g3_state_dict():
    out = {}
    for seed in (42, 123):
        for reg in (False, True):
            cls = ChromoformerRegressor if reg else ChromoformerClassifier
            m = cls(7, 128, 128, *kws(), seed=seed)
            sd = m.state_dict()
            P = orc.init_params(None, seed=seed, regression=reg)
            assert list(P.keys()) == list(sd.keys())
            for k in sd:
                assert torch.equal(P[k], sd[k]), k
            out["seed%d_%s" % (seed, "reg" if reg else "clf")] = {
                "keys": list(sd.keys()),
                "shapes": [list(v.shape) for v in sd.values()],
                "checksums": [checksum(v) for v in sd.values()],
            }
    m = ChromoformerClassifier(7, 128, 128, *kws(), seed=42)
    out["n_params"] = int(sum(v.numel() for v in m.state_dict().values()))
    with open(os.path.join(HERE, "state_dict.json"), "w") as f:
        json.dump(out, f)
    print("G3 ok: oracle init bit-identical to reference for seeds 42/123, clf+reg; n_params", out["n_params"])


def g2_kat():
    m0 = Chromoformer()
    mc = ChromoformerClassifier()
    mr = ChromoformerRegressor()
    bsz, S = 8, 8
    Ls = {2000: 20, 500: 80, 100: 400}
    xp = {b: torch.randn([bsz, 1, Ls[b], 7]) for b in BINS}
    xc = {b: torch.randn([bsz, S, Ls[b], 7]) for b in BINS}
    mp = {b: torch.randn([bsz, 1, 1, Ls[b], Ls[b]]).bool() for b in BINS}
    mcm = {b: torch.randn([bsz, S, 1, Ls[b], Ls[b]]).bool() for b in BINS}
    im = {b: torch.randn([bsz, 1, 1 + S, 1 + S]).bool() for b in BINS}
    fr = torch.randn([bsz, 1 + S, 1 + S])
    batch = {"promoter_feats": xp, "promoter_pad_masks": mp, "pcre_feats": xc, "pcre_pad_masks": mcm,
             "interaction_masks": im, "interaction_freq": fr}
    with torch.no_grad():
        o0 = m0(xp[2000], mp[2000], xc[2000], mcm[2000], im[2000], xp[500], mp[500], xc[500], mcm[500], im[500],
                xp[100], mp[100], xc[100], mcm[100], im[100], fr)
        oc = ref_call(mc, batch)
        orr = ref_call(mr, batch)
        print("G2 sums", float(o0.sum()), float(oc.sum()), float(orr.sum()))
        assert abs(float(o0.sum()) - (-3.1917)) < 5e-4 and abs(float(oc.sum()) - (-3.1917)) < 5e-4
        assert abs(float(orr.sum()) - (-0.1900)) < 5e-4
        assert torch.equal(o0, oc)
        # oracle on the same inputs / weights
        Pc = orc.init_params(None, 42, False)
        Pr = orc.init_params(None, 42, True)
        dc = (orc.forward(Pc, batch) - oc).abs().max().item()
        dr = (orc.forward(Pr, batch) - orr).abs().max().item()
        print("G2 oracle-vs-reference max|d|:", dc, dr)
        assert dc < 1e-6 and dr < 1e-6
    flat_save(os.path.join(HERE, "kat.npz"), batch,
              {"logits_clf": oc.numpy(), "logits_reg": orr.numpy()})


def demo_dataset():
    meta = os.path.join(REF, "demo/demo_meta.csv")
    genes = pd.read_csv(meta).gene_id.tolist()
    return ChromoformerDataset(meta, os.path.join(REF, "demo/demo_data"), genes), pd.read_csv(meta)


def g1_demo():
    ds, meta = demo_dataset()
    items = [ds[i] for i in range(len(ds))]
    torch.manual_seed(123)
    model = ChromoformerClassifier(7, 128, 128, *kws(), seed=123)
    model.eval()
    P = orc.init_params(None, 123, False)
    logits = []
    with torch.no_grad():
        for i in range(0, 100, 25):
            logits.append(ref_call(model, collate(items[i:i + 25])))
    logits = torch.cat(logits)
    pred = torch.sigmoid(logits)[:, 1].numpy()
    committed = pd.read_csv(os.path.join(REF, "demo/random_prediction.out")).prediction.values
    d = np.abs(pred - committed).max()
    print("G1 reference-here vs demo/random_prediction.out max|d| =", d)
    assert d < 5e-7
    n_part = meta.neighbors.fillna("").apply(lambda s: 0 if s == "" else len(s.split(";"))).values
    strand = meta.strand.values
    want = [(0, "+"), (1, "-"), (5, "+"), (8, "-"), (8, "+"), (3, "-")]
    pick = []
    for n, st in want:
        for i in range(100):
            if n_part[i] == n and strand[i] == st and i not in pick:
                pick.append(i)
                break
    print("G1 subset rows", pick, [(int(n_part[i]), strand[i]) for i in pick])
    sub = collate([items[i] for i in pick])
    with torch.no_grad():
        ref_sub = ref_call(model, sub)
        o_sub, stages = orc.forward(P, sub, return_stages=True)
        assert (ref_sub - logits[pick]).abs().max() < 2e-6
        d = (o_sub - ref_sub).abs().max().item()
        print("G1 oracle-vs-reference (subset) max|d| =", d)
        assert d < 1e-6
        # all 100 genes through the oracle
        o_all = torch.cat([orc.forward(P, collate(items[i:i + 25])) for i in range(0, 100, 25)])
        d = (o_all - logits).abs().max().item()
        print("G1 oracle-vs-reference (100 genes) max|d| =", d)
        assert d < 1e-6
        # G7 stage intermediates from the reference's own submodules
        g7 = {}
        for b in BINS:
            full, tss = model.embed[str(b)](sub["promoter_feats"][b], sub["promoter_pad_masks"][b])
            pw = model.pairwise_interaction[str(b)](full, sub["pcre_feats"][b], sub["pcre_pad_masks"][b])
            x_in = torch.cat([tss, pw], dim=1)
            x_out = model.regulation[str(b)](x_in, sub["interaction_masks"][b], sub["interaction_freq"])
            g7["g7.embed_tss.%d" % b] = tss.numpy()
            g7["g7.pairwise.%d" % b] = pw.numpy()
            g7["g7.regulation_row0.%d" % b] = x_out[:, 0].numpy()
            for k in ("embed_tss", "pairwise", "regulation_row0"):
                dd = (stages["%s.%d" % (k, b)].reshape(-1) - torch.from_numpy(g7["g7.%s.%d" % (k, b)]).reshape(-1)).abs().max().item()
                assert dd < 2e-6, (k, b, dd)
        # G8 structural exactness: dead promoter rows and dummy pCRE features never reach the logits
        poisoned = {k: (dict(v) if isinstance(v, dict) else v) for k, v in sub.items()}
        poisoned["pcre_feats"] = {b: t.clone() for b, t in sub["pcre_feats"].items()}
        for gi, i in enumerate(pick):
            for b in BINS:
                poisoned["pcre_feats"][b][gi, n_part[i]:] = 7.5
        d8 = (ref_call(model, poisoned) - ref_sub).abs().max().item()
        print("G8 dummy-pCRE poison -> logits max|d| =", d8)
        assert d8 == 0.0
    flat_save(os.path.join(HERE, "demo_subset.npz"), sub,
              dict(g7, logits=ref_sub.numpy(), rows=np.array(pick), n_partners=n_part[pick],
                   all_logits=logits.numpy()))
    return sub, pick


def g4_train_step(sub):
    idx = [1, 2, 3, 4]
    batch = {k: ({b: t[idx] for b, t in v.items()} if isinstance(v, dict) else v[idx]) for k, v in sub.items()}
    batch["label"] = torch.tensor([1, 0, 1, 0])
    out = {}
    for reg in (False, True):
        tag = "reg" if reg else "clf"
        cls = ChromoformerRegressor if reg else ChromoformerClassifier
        model = cls(7, 128, 128, *kws(), seed=42)
        crit = torch.nn.MSELoss() if reg else torch.nn.CrossEntropyLoss()
        opt = torch.optim.AdamW(model.parameters(), lr=float("3e-5"))
        opt.zero_grad()
        opt.step()
        label = torch.tensor([2.5, 0.25, 4.0, 0.0]).view(-1, 1) if reg else batch["label"]
        opt.zero_grad()
        logits = ref_call(model, batch)
        loss = crit(logits, label)
        loss.backward()
        grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}
        opt.step()
        # oracle, same step
        P = orc.init_params(None, 42, reg)
        for t in P.values():
            t.requires_grad_(True)
        oopt = orc.make_optimizer(P, "3e-5")
        ob = dict(batch)
        ob["label"] = label.view(-1) if reg else label
        oloss, ologits = orc.train_step(P, oopt, ob, regression=reg)
        assert abs(float(oloss) - float(loss)) < 1e-6, (float(oloss), float(loss))
        none_names = [k for k, g in grads.items() if g is None]
        assert sorted(none_names) == sorted(k for k in P if orc.never_trained(k)), "never-trained set differs"
        worst = 0.0
        for k, g in grads.items():
            og = P[k].grad
            if g is None:
                assert og is None or float(og.abs().max()) == 0.0, k
                continue
            scale = float(g.abs().max()) + 1e-12
            worst = max(worst, float((og - g).abs().max()) / scale)
        print("G4 [%s] loss %.6f; oracle grads vs reference: worst rel-to-max err %.2e" % (tag, float(loss), worst))
        assert worst < 5e-4
        sd = model.state_dict()
        dpar = max(float((P[k].detach() - sd[k]).abs().max()) for k in sd)
        print("G4 [%s] post-AdamW params oracle vs reference max|d| = %.2e" % (tag, dpar))
        assert dpar < 1e-7
        names = list(grads.keys())
        out["%s.loss" % tag] = np.float64(loss.item())
        out["%s.logits" % tag] = logits.detach().numpy()
        out["%s.label" % tag] = label.numpy()
        out["%s.grad_checksums" % tag] = np.array([[0, 0, 0] if grads[k] is None else checksum(grads[k]) for k in names])
        out["%s.grad_is_none" % tag] = np.array([grads[k] is None for k in names])
        out["%s.param_checksums_after" % tag] = np.array([checksum(sd[k]) for k in names])
        for k in names:  # full small tensors
            if grads[k] is not None and grads[k].numel() <= 1024:
                out["%s.grad.%s" % (tag, k)] = grads[k].numpy()
        # a few big ones in full as well
        for k in ("embed.100.transformer.layers.0.self_att.att.weight",
                  "pairwise_interaction.100.transformer.layers.1.self_att.c_att.weight",
                  "pairwise_interaction.500.lin_proj_p.weight",
                  "regulation.2000.transformer.layers.0.self_att.att.weight",
                  "regulation.100.transformer.layers.5.ff.l2.weight", "fc_head.0.weight"):
            out["%s.grad.%s" % (tag, k)] = grads[k].numpy()
        osd = opt.state_dict()
        out["%s.opt_state_keys" % tag] = np.array(sorted(osd["state"].keys()))
        if not reg:
            pg = dict(osd["param_groups"][0])
            pg.pop("params")
            with open(os.path.join(HERE, "optimizer_param_group.json"), "w") as f:
                json.dump({k: (list(v) if isinstance(v, tuple) else v) for k, v in pg.items()}, f)
            st0 = osd["state"][sorted(osd["state"].keys())[0]]
            out["clf.opt_state_entry_fields"] = np.array(sorted(st0.keys()))
            out["clf.opt_step_dtype"] = np.array(str(st0["step"].dtype))
    out["names"] = np.array(names)
    out["rows_in_demo_subset"] = np.array(idx)
    np.savez_compressed(os.path.join(HERE, "train_step.npz"), **out)
    print("wrote train_step.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "train_step.npz")) / 1024))


def g5_dataset():
    rng = np.random.default_rng(7)
    tmp = tempfile.mkdtemp()

    def synth(length):
        lam = np.abs(np.cumsum(rng.normal(0, 0.05, size=(7, length)), axis=1)) * 0.4
        x = rng.poisson(lam).astype(np.float16)
        x[:, rng.random(length) < 0.3] = 0
        return x

    regions = {}
    genes = [
        # gene, chrom, tss, strand, label, expr, [(start, len)], scores
        ("G_PLUS", "chrA", 500000, "+", 1, 3.25, [(620000, 5900), (300100, 1833), (540000, 12345)], [2.5, 1.9, 1.6]),
        ("G_MINUS", "chrA", 900000, "-", 0, 0.5, [(990000, 40000), (700000, 2001), (880000, 100), (860000, 7777),
                                                    (840000, 3999), (820000, 25000), (800000, 4100), (780000, 9050)],
         [2.9, 2.2, 2.0, 1.9, 1.8, 1.7, 1.6, 1.55]),
        ("G_NONE", "chrB", 100000, "-", 1, 7.0, [], []),
    ]
    rows = []
    for gid, chrom, tss, strand, label, expr, pc, sc in genes:
        regions["%s:%d-%d" % (chrom, tss - 20000, tss + 20000)] = synth(40000)
        names = []
        for st, ln in pc:
            nm = "%s:%d-%d" % (chrom, st, st + ln)
            regions[nm] = synth(ln)
            names.append(nm)
        rows.append(dict(gene_id=gid, expression=expr, eid="E000", label=label, chrom=chrom, start=tss, end=tss + 1,
                         strand=strand, split=1, neighbors=";".join(names) if names else np.nan,
                         scores=";".join(str(s) for s in sc) if sc else np.nan))
    for nm, arr in regions.items():
        np.save(os.path.join(tmp, nm + ".npy"), arr)
    meta = os.path.join(tmp, "meta.csv")
    pd.DataFrame(rows).to_csv(meta, index=False)
    out = {"meta_csv": np.array(open(meta).read())}
    for nm, arr in regions.items():
        out["raw." + nm] = arr
    for reg in (False, True):
        for w_prom in (40000, 10000):
            ds = ChromoformerDataset(meta, tmp, [r["gene_id"] for r in rows], w_prom=w_prom, regression=reg)
            for gi, r in enumerate(rows):
                it = ds[gi]
                pcs = [] if not isinstance(r["neighbors"], str) else r["neighbors"].split(";")
                pcs = [(p.split(":")[0], int(p.split(":")[1].split("-")[0]), int(p.split(":")[1].split("-")[1])) for p in pcs]
                scs = [] if not isinstance(r["scores"], str) else [float(s) for s in r["scores"].split(";")]
                lab = np.log2(r["expression"] + 1) if reg else r["label"]
                mine = dso.gene_item(lambda c, s, e: regions["%s:%d-%d" % (c, s, e)], (r["chrom"], r["start"]),
                                     r["strand"], pcs, scs, lab, w_prom=w_prom, regression=reg)
                tag = "item.%s.w%d.%s" % ("reg" if reg else "clf", w_prom, r["gene_id"])
                for k, v in it.items():
                    if isinstance(v, dict):
                        for b, t in v.items():
                            o = mine[k][b]
                            assert o.shape == t.shape and o.dtype == t.dtype, (k, b, o.shape, t.shape)
                            if t.dtype == torch.bool:
                                assert torch.equal(o, t), (tag, k, b)
                            else:
                                assert (o - t).abs().max() < 2e-6, (tag, k, b, (o - t).abs().max())
                            if not reg:
                                out["%s.%s.%d" % (tag, k, b)] = t.numpy()
                    else:
                        assert torch.allclose(mine[k].float(), v.float(), atol=1e-6), (tag, k)
                        assert mine[k].dtype == v.dtype
                        out["%s.%s" % (tag, k)] = v.numpy()
    np.savez_compressed(os.path.join(HERE, "dataset.npz"), **out)
    print("G5 ok; wrote dataset.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "dataset.npz")) / 1024))


def g6_training_run():
    """Run the reference's own `python -m chromoformer.train` (CPU, wandb / .cuda() stubbed) on the
    deterministic synthetic dataset of tests/synth_data.py and keep what its checkpoint holds."""
    import runpy
    import yaml
    from tests.synth_data import make_dataset
    tmp = tempfile.mkdtemp()
    meta = make_dataset(os.path.join(tmp, "npy"), n_genes=48, seed=2024)
    cfg = yaml.safe_load(open(os.path.join(REF, "chromoformer/configs/default.yaml")))
    cfg["bsz"] = 8
    cfg["num_epoch"] = 3
    cfg_path = os.path.join(tmp, "cfg.yaml")
    yaml.safe_dump(cfg, open(cfg_path, "w"))
    wb = sys.modules["wandb"]
    wb.init = lambda *a, **k: None
    wb.log = lambda *a, **k: None
    wb.config = types.SimpleNamespace(update=lambda *a, **k: None)
    wb.summary = types.SimpleNamespace(update=lambda *a, **k: None)
    import torch.utils.data as tud
    real_loader = tud.DataLoader

    def single_process_loader(*a, **k):      # same sampling / RNG consumption, no worker processes
        k["num_workers"] = 0
        return real_loader(*a, **k)

    torch.utils.data.DataLoader = single_process_loader
    torch.autograd.set_detect_anomaly(False)
    out = {}
    for reg in (False, True):
        ck = os.path.join(tmp, "ck_%d.pt" % reg)
        argv = ["train", "-o", ck, "-c", cfg_path, "--exp-id", "g6", "-m", meta, "-d", os.path.join(tmp, "npy"), "--fold", "0"]
        if reg:
            argv.append("--regression")
        old = sys.argv
        sys.argv = argv
        try:
            runpy.run_module("chromoformer.train", run_name="__main__")
        finally:
            sys.argv = old
            torch.autograd.set_detect_anomaly(False)
        c = torch.load(ck, map_location="cpu", weights_only=False)
        tag = "reg" if reg else "clf"
        out[tag + ".epoch"] = np.int64(c["epoch"])
        out[tag + ".last_val_loss"] = np.float64(float(c["last_val_loss"]))
        out[tag + ".last_val_metric"] = np.float64(c["last_val_r2" if reg else "last_val_auc"])
        out[tag + ".val_score"] = np.asarray(c["val_score"])
        out[tag + ".val_label"] = np.asarray(c["val_label"])
        out[tag + ".lr"] = np.float64(c["optimizer"]["param_groups"][0]["lr"])
        out[tag + ".opt_n_state"] = np.int64(len(c["optimizer"]["state"]))
        out[tag + ".opt_step"] = np.float64(float(next(iter(c["optimizer"]["state"].values()))["step"]))
        out[tag + ".ckpt_keys"] = np.array(list(c.keys()))
        out[tag + ".param_checksums"] = np.array([checksum(v) for v in c["net"].values()])
        out[tag + ".val_score_dtype"] = np.array(str(np.asarray(c["val_score"]).dtype))
        out[tag + ".val_label_dtype"] = np.array(str(np.asarray(c["val_label"]).dtype))
        print("G6 [%s] epoch %d val_loss %.5f metric %.4f lr %.4e" % (tag, c["epoch"], float(c["last_val_loss"]),
              out[tag + ".last_val_metric"], out[tag + ".lr"]))
    torch.utils.data.DataLoader = real_loader
    np.savez_compressed(os.path.join(HERE, "train_run.npz"), **out)
    print("wrote train_run.npz")


if __name__ == "__main__":
    which = sys.argv[1:] or ["g3", "g2", "g1", "g5", "g6"]
    if "g6" in which:
        g6_training_run()
    if "g3" in which:
        g3_state_dict()
    if "g2" in which:
        g2_kat()
    if "g1" in which:
        sub, _ = g1_demo()
        g4_train_step(sub)
    if "g5" in which:
        g5_dataset()
