"""Parity of the HIP hot path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Tolerances: logits 1e-4 absolute (BASELINE.json north_star), gradients
1e-3 of each tensor's max-abs, AdamW-updated parameters 1e-6."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from oracle import chromoformer_oracle as orc
from tests.helpers import GOLDEN, load_npz_batch, take

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-4
GRAD_TOL = 1e-3


def _model(reg=False, seed=42, max_batch=8):
    from chromoformer_amd import ChromoformerClassifier, ChromoformerRegressor
    return (ChromoformerRegressor if reg else ChromoformerClassifier)(seed=seed, max_batch=max_batch).cuda(0)


def _call(model, batch):
    return model(batch["promoter_feats"], batch["promoter_pad_masks"], batch["pcre_feats"], batch["pcre_pad_masks"],
                 batch["interaction_masks"], batch["interaction_freq"])


# ------------------------------------------------------------------ standalone operators
def test_ops_linear_wgrad_dgrad():
    from chromoformer_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    for M, N, K in ((37, 128, 128), (576, 1024, 128), (64, 128, 384), (5, 256, 256)):
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) * 0.1
        b = torch.randn(N, generator=g)
        dY = torch.randn(M, N, generator=g)
        Ad, Wd, bd, dYd = (t.cuda() for t in (A, W, b, dY))
        Cd = torch.empty(M, N, device="cuda")
        _lib.check(L.cf_op_linear(Ad.data_ptr(), Wd.data_ptr(), bd.data_ptr(), Cd.data_ptr(), M, N, K, 1, None))
        ref = torch.relu(A.double() @ W.double().t() + b.double())
        assert (Cd.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
        dW = torch.empty(N, K, device="cuda")
        _lib.check(L.cf_op_wgrad(dYd.data_ptr(), Ad.data_ptr(), dW.data_ptr(), M, N, K, None))
        ref = dY.double().t() @ A.double()
        assert (dW.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())
        dX = torch.empty(M, K, device="cuda")
        _lib.check(L.cf_op_dgrad(dYd.data_ptr(), Wd.data_ptr(), dX.data_ptr(), M, N, K, None))
        ref = dY.double() @ W.double()
        assert (dX.cpu().double() - ref).abs().max() < 2e-5 * max(1.0, ref.abs().max().item())


# ------------------------------------------------------------------ forward
def test_g1_demo_subset_logits_and_stages():
    batch, ex = load_npz_batch("demo_subset.npz")
    model = _model(seed=123)
    with torch.no_grad():
        out = _call(model, batch)
    assert np.abs(out.cpu().numpy() - ex["logits"]).max() < LOGIT_TOL
    T = 9
    for r, b in enumerate((2000, 500, 100)):
        x0 = model.debug_buffer("R%d.x0" % r).cpu().view(-1, T, 128)[: out.shape[0]]
        assert np.abs(x0[:, 0].numpy() - ex["g7.embed_tss.%d" % b][:, 0]).max() < LOGIT_TOL
        n_part = ex["n_partners"]
        for g in range(out.shape[0]):   # dummy pCRE slots are dead values; compare the live ones
            k = int(n_part[g])
            assert np.abs(x0[g, 1:1 + k].numpy() - ex["g7.pairwise.%d" % b][g, :k]).max(initial=0) < LOGIT_TOL
        xl = model.debug_buffer("R%d.x6" % r).cpu().view(-1, T, 128)[: out.shape[0]]
        assert np.abs(xl[:, 0].numpy() - ex["g7.regulation_row0.%d" % b]).max() < LOGIT_TOL


def test_g2_known_answers_fully_masked():
    batch, ex = load_npz_batch("kat.npz")
    for reg in (False, True):
        model = _model(reg=reg, seed=42)
        with torch.no_grad():
            out = _call(model, batch).cpu()
        ref = ex["logits_reg" if reg else "logits_clf"]
        assert np.abs(out.numpy() - ref).max() < LOGIT_TOL
        assert abs(float(out.sum()) - (-0.1900 if reg else -3.1917)) < 1e-3    # net.py:558-568


def test_legacy_class_equals_classifier():
    """net.py:156-270 / the smoke block net.py:431-568: `Chromoformer()` (16 positional arguments, embed2000 / pw_int2000 /
    reg2000 module names) and ChromoformerClassifier give the same output for the same seed."""
    from chromoformer_amd import Chromoformer
    batch, ex = load_npz_batch("kat.npz")
    legacy = Chromoformer(seed=42, max_batch=16).cuda(0)
    args = []
    for b in (2000, 500, 100):
        args += [batch["promoter_feats"][b], batch["promoter_pad_masks"][b], batch["pcre_feats"][b], batch["pcre_pad_masks"][b],
                 batch["interaction_masks"][b]]
    with torch.no_grad():
        out = legacy(*args, batch["interaction_freq"]).cpu()
    assert np.abs(out.numpy() - ex["logits_clf"]).max() < LOGIT_TOL and abs(float(out.sum()) - (-3.1917)) < 1e-3
    sd = legacy.state_dict()
    keys = list(sd)
    assert keys[0] == "embed2000.lin_proj.weight" and "pw_int500.lin_proj_p.weight" in sd and "reg100.transformer.layers.5.ff.l2.bias" in sd
    clf = _model(seed=7)
    legacy.load_state_dict(type(sd)((k, v) for k, v in zip(keys, clf.state_dict().values())))
    with torch.no_grad():
        assert torch.equal(legacy(*args, batch["interaction_freq"]), _call(clf, batch))


def test_compact_mask_rows_equal_full_masks():
    batch = orc.synthetic_batch(4, seed=9, regime="realistic")
    model = _model()
    with torch.no_grad():
        full = _call(model, batch).cpu()
        compact = dict(batch)
        compact["promoter_pad_masks"] = {b: m[:, 0, 0, m.shape[-1] // 2, :].contiguous() for b, m in batch["promoter_pad_masks"].items()}
        compact["pcre_pad_masks"] = {b: m[:, :, 0, m.shape[-1] // 2, :].contiguous() for b, m in batch["pcre_pad_masks"].items()}
        out = _call(model, compact).cpu()
    assert torch.equal(full, out)


def test_forward_vs_oracle_random_weights_and_ragged_batches():
    P = orc.init_params(None, 7, False)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        for v in P.values():
            v.add_(0.05 * torch.randn(v.shape, generator=g))
    model = _model(max_batch=19)
    model.load_state_dict(P)
    for B, regime in ((1, "dense"), (19, "realistic"), (16, "dense")):
        batch = orc.synthetic_batch(B, seed=B, regime=regime)
        with torch.no_grad():
            ref = orc.forward(P, batch)
            out = _call(model, batch).cpu()
        assert (out - ref).abs().max() < LOGIT_TOL, (B, regime)


# ------------------------------------------------------------------ backward / optimiser
def _perturbed(seed, reg):
    P = orc.init_params(None, seed, reg)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for v in P.values():
            v.add_(0.05 * torch.randn(v.shape, generator=g))
    return P


@pytest.mark.parametrize("reg", [False, True])
def test_gradients_vs_oracle_autograd(reg):
    B = 6
    batch = orc.synthetic_batch(B, seed=21, regime="realistic", regression=reg)
    P = _perturbed(42, reg)
    model = _model(reg=reg, max_batch=B)
    model.load_state_dict(P)
    for t in P.values():
        t.requires_grad_(True)
    ref_logits = orc.forward(P, batch)
    ref_loss = orc.loss_fn(ref_logits, batch["label"], reg)
    ref_loss.backward()
    logits, loss = model.forward_backward(model.pack_batch(batch), batch["label"])
    assert (logits.cpu() - ref_logits.detach()).abs().max() < LOGIT_TOL
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    model._publish_grads()
    named = dict(model.named_parameters())
    for k, v in P.items():
        if orc.never_trained(k):
            assert named[k].grad is None
            continue
        ref = v.grad
        err = (named[k].grad.cpu() - ref).abs().max().item()
        assert err <= GRAD_TOL * ref.abs().max().item() + 1e-9, (k, err, ref.abs().max().item())


def test_drop_in_autograd_path_matches_fused_path():
    B = 4
    batch = orc.synthetic_batch(B, seed=5, regime="realistic")
    model = _model(max_batch=B)
    out = _call(model, batch)
    loss = torch.nn.CrossEntropyLoss()(out, batch["label"].cuda())
    loss.backward()
    g1 = model.active_grads().clone()
    _, loss2 = model.forward_backward(model.pack_batch(batch), batch["label"])
    assert abs(loss.item() - loss2.item()) < 1e-6
    g2 = model.active_grads()
    assert (g1 - g2).abs().max().item() <= 1e-6 * max(1.0, g2.abs().max().item())
    assert sum(p.grad is not None for p in model.parameters()) == 334


@pytest.mark.parametrize("reg", [False, True])
def test_g4_train_step_golden(reg):
    sub, _ = load_npz_batch("demo_subset.npz")
    z = np.load(os.path.join(GOLDEN, "train_step.npz"))
    batch = take(sub, list(z["rows_in_demo_subset"]))
    tag = "reg" if reg else "clf"
    names = list(z["names"])
    model = _model(reg=reg, seed=42, max_batch=4)
    label = torch.from_numpy(z[tag + ".label"]).view(-1)
    logits, loss = model.train_step(model.pack_batch(batch), label, float("3e-5"))
    assert abs(loss.item() - float(z[tag + ".loss"])) < 1e-4
    assert np.abs(logits.cpu().numpy() - z[tag + ".logits"]).max() < LOGIT_TOL
    model._publish_grads()
    named = dict(model.named_parameters())
    for i, k in enumerate(names):
        if z[tag + ".grad_is_none"][i]:
            continue
        key = tag + ".grad." + k
        if key in z.files:
            ref = z[key]
            err = np.abs(named[k].grad.cpu().numpy() - ref).max()
            assert err <= GRAD_TOL * np.abs(ref).max() + 1e-9, (k, err)
    # post-AdamW parameters: oracle does the same step on the CPU
    P = orc.init_params(None, 42, reg)
    for t in P.values():
        t.requires_grad_(True)
    opt = orc.make_optimizer(P, "3e-5")
    b = dict(batch)
    b["label"] = label
    orc.train_step(P, opt, b, regression=reg)
    # the first AdamW update is u = lr * g / (|g| + eps) with du/dg = lr * eps / (|g| + eps)^2: where |g| is far
    # above eps = 1e-8 the update is insensitive to rounding noise in g (checked to 2e-7); where |g| ~ eps the
    # noise of g (~1e-9, in the oracle as well) is amplified up to a fraction of lr = 3e-5 (bounded by 2e-5)
    sd = model.state_dict()
    for k in P:
        d = (sd[k].cpu() - P[k].detach()).abs()
        if P[k].grad is None:
            assert d.max().item() == 0.0, k
            continue
        big = P[k].grad.abs() > 1e-6
        assert d[big].max().item() < 2e-7 if big.any() else True, k
        assert d.max().item() < 2e-5, k


def test_three_adamw_steps_track_oracle():
    B = 4
    model = _model(max_batch=B)
    P = orc.init_params(None, 42, False)
    for t in P.values():
        t.requires_grad_(True)
    opt = orc.make_optimizer(P, 1e-3)
    for s in range(3):
        batch = orc.synthetic_batch(B, seed=100 + s, regime="realistic")
        ref_loss, _ = orc.train_step(P, opt, batch)
        _, loss = model.train_step(model.pack_batch(batch), batch["label"], 1e-3)
        assert abs(loss.item() - ref_loss.item()) < 2e-4
    sd = model.state_dict()
    worst = max((sd[k].cpu() - P[k].detach()).abs().max().item() for k in P)
    assert worst < 2e-4, worst     # lr 1e-3 * 3 steps = 3e-3 of movement; sign-level agreement of the updates


# ------------------------------------------------------------------ full-size properties (B = 64)
def test_full_batch_determinism_and_batch_linearity():
    B = 64
    batch = orc.synthetic_batch(B, seed=1234, regime="dense")
    model = _model(max_batch=B)
    packed = model.pack_batch(batch)
    l1, _ = model.forward_backward(packed, batch["label"])
    g1 = model.active_grads().clone()
    l2, _ = model.forward_backward(packed, batch["label"])
    assert torch.equal(l1, l2) and torch.equal(g1, model.active_grads())      # run-to-run bit stability
    halves = []
    for idx in (list(range(0, 32)), list(range(32, 64))):
        hb = take(batch, idx)
        lg, _ = model.forward_backward(model.pack_batch(hb), hb["label"])
        assert (lg - l1[idx]).abs().max().item() < 1e-5                        # genes are independent units
        halves.append(model.active_grads().clone())
    avg = 0.5 * (halves[0] + halves[1])                                        # what a 2-rank all-reduce(avg) yields
    assert (avg - g1).abs().max().item() <= 2e-5 * g1.abs().max().item() + 1e-8
    with torch.no_grad():
        ref = orc.forward(orc.init_params(None, 42, False), take(batch, list(range(8))))
    assert (l1[:8].cpu() - ref).abs().max() < LOGIT_TOL


def test_library_fails_loudly_on_bad_input():
    model = _model(max_batch=2)
    batch = orc.synthetic_batch(3, seed=1)
    with pytest.raises(ValueError):
        model.pack_batch(batch)
