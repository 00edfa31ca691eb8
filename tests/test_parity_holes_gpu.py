"""Round-3 parity cases the earlier suites left open ("close the parity holes"):

  (a) an fp64 REFEREE at the benchmark's own batch size: the oracle evaluated in float64 is the truth, the fp32 oracle and the
      HIP path are two fp32 evaluations of it -- per tensor the HIP error must not exceed twice the fp32 oracle's own error
      (or 2e-5 of the tensor's norm where the fp32 oracle happens to land closer than that).  This is the evidence behind the
      1e-3 Frobenius tolerance of tests/test_full_size_gpu.py: the distance to the fp32 oracle is fp32 noise of BOTH sides;
  (b) the regressor (MSE head, n_out = 1) at bsz 64;
  (c) promoter padding (w_prom = 10000 < w_max, data.py:136-162): the reference's own `__getitem__` tensors of golden G5,
      5-d masks and all, through the model's forward / backward against the oracle;
  (d) the stress shape at its real size (N = 128 x 17 sequences, L = 800): run-to-run determinism of forward + backward and
      an oracle check (fp64) on four sampled sequences of that run."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import chromoformer_oracle as orc
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
B = 64


def _hip(Model, batch, P):
    model = Model(seed=42, max_batch=batch["interaction_freq"].shape[0]).cuda(0)
    model.load_state_dict(P)
    logits, loss = model.forward_backward(model.pack_batch(batch), batch["label"])
    torch.cuda.synchronize()
    model._publish_grads()
    return logits.cpu().clone(), float(loss), {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}


def _oracle(P, batch, regression, dtype):
    Pr = {k: v.detach().to(dtype).clone().requires_grad_(True) for k, v in P.items()}
    b = {k: ({kk: (vv.to(dtype) if vv.is_floating_point() else vv) for kk, vv in v.items()} if isinstance(v, dict) else v) for k, v in batch.items()}
    b["interaction_freq"] = batch["interaction_freq"].to(dtype)
    logits = orc.forward(Pr, b)
    if regression:
        loss = F.mse_loss(logits, batch["label"].view(-1, 1).to(dtype))
    else:
        loss = F.cross_entropy(logits, batch["label"].long())
    loss.backward()
    return logits.detach(), float(loss), {k: v.grad for k, v in Pr.items() if v.grad is not None and not orc.never_trained(k)}


def _perturbed(regression):
    P = orc.init_params(None, 42, regression)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for v in P.values():
            v.add_(0.02 * torch.randn(v.shape, generator=g))
    return P


@pytest.mark.parametrize("regression", [False, True], ids=["classifier", "regressor"])
def test_bsz64_against_an_fp64_referee(regression):
    from chromoformer_amd import ChromoformerClassifier, ChromoformerRegressor
    batch = orc.synthetic_batch(B, seed=2024, regime="dense", regression=regression)
    P = _perturbed(regression)
    l64, loss64, g64 = _oracle(P, batch, regression, torch.float64)
    l32, loss32, g32 = _oracle(P, batch, regression, torch.float32)
    lh, lossh, gh = _hip(ChromoformerRegressor if regression else ChromoformerClassifier, batch, P)
    # forward: both fp32 evaluations sit within 1e-4 of the truth; the HIP path is not further from it than twice the oracle
    e32, eh = (l32.double() - l64).abs().max().item(), (lh.double() - l64).abs().max().item()
    assert eh < 1e-4 and eh <= max(2 * e32, 2e-6), (eh, e32)
    assert abs(lossh - loss64) <= max(2 * abs(loss32 - loss64), 2e-6 * max(1.0, abs(loss64)))
    assert set(gh) == set(g64) and len(g64) == 334
    worst = (0.0, None)
    for k, ref in g64.items():
        n = ref.norm().item()
        err_h, err_32 = (gh[k].double() - ref).norm().item(), (g32[k].double() - ref).norm().item()
        assert err_h <= max(2 * err_32, 2e-5 * n) + 1e-12, (k, err_h / max(n, 1e-30), err_32 / max(n, 1e-30))
        worst = max(worst, (err_h / max(err_32, 1e-30), k))
        # and the absolute statement: relative Frobenius error against the TRUTH below 1e-3 for every tensor
        assert err_h <= 1e-3 * n + 1e-12, (k, err_h / n)
    print("worst err(HIP) / err(fp32 oracle) against the fp64 referee: %.2f (%s)" % worst)


def _g5_batch(tag, w):
    z = np.load(os.path.join(GOLDEN, "dataset.npz"))
    genes = ["G_PLUS", "G_MINUS", "G_NONE"]
    batch = {k: {} for k in ("promoter_feats", "promoter_pad_masks", "pcre_feats", "pcre_pad_masks", "interaction_masks")}
    for key in batch:
        for b in (2000, 500, 100):
            batch[key][b] = torch.stack([torch.from_numpy(z["item.%s.w%d.%s.%s.%d" % (tag, w, g, key, b)]) for g in genes])      # default collate
    batch["interaction_freq"] = torch.stack([torch.from_numpy(z["item.%s.w%d.%s.interaction_freq" % (tag, w, g)]) for g in genes])
    batch["label"] = torch.stack([torch.from_numpy(z["item.%s.w%d.%s.label" % (tag, w, g)]) for g in genes])
    return batch


@pytest.mark.parametrize("w_prom", [10000, 40000])
def test_reference_items_with_promoter_padding_forward_and_backward(w_prom):
    """G5 items exactly as the reference's DataLoader collates them (5-d bool masks): at w_prom = 10000 the promoter covers
    5 / 20 / 100 of the 20 / 80 / 400 bins, centred -- the Embedding centre row attends over the valid promoter bins only."""
    from chromoformer_amd import ChromoformerClassifier
    batch = _g5_batch("clf", w_prom)
    L = 400
    valid = (~batch["promoter_pad_masks"][100][:, 0, 0, L // 2]).sum(1)
    assert valid.tolist() == [w_prom // 100] * 3 and batch["promoter_pad_masks"][100].shape == (3, 1, 1, 400, 400)
    P = _perturbed(False)
    l32, loss32, g32 = _oracle(P, batch, False, torch.float32)
    model = ChromoformerClassifier(seed=42, max_batch=3).cuda(0)
    model.load_state_dict(P)
    # (1) the reference's six-argument forward, device tensors in the reference layout
    dev = {k: ({b: t.cuda() for b, t in v.items()} if isinstance(v, dict) else v.cuda()) for k, v in batch.items()}
    with torch.no_grad():
        out = model(dev["promoter_feats"], dev["promoter_pad_masks"], dev["pcre_feats"], dev["pcre_pad_masks"], dev["interaction_masks"],
                    dev["interaction_freq"])
    torch.cuda.synchronize()
    assert (out.cpu() - l32).abs().max() < 1e-4
    # (2) loss and all gradients
    lh, lossh, gh = _hip(ChromoformerClassifier, batch, P)
    assert (lh - l32).abs().max() < 1e-4 and abs(lossh - loss32) < 1e-5 * max(1.0, abs(loss32))
    for k, ref in g32.items():
        assert (gh[k] - ref).abs().max() <= 1e-3 * ref.abs().max() + 1e-9, (k, (gh[k] - ref).abs().max().item(), ref.abs().max().item())


def test_stress_shape_at_full_size_is_deterministic_and_matches_the_oracle_on_sampled_sequences():
    """BASELINE configs[3]: N = 128 genes x 17 regions, 2 heads, L = 800, dh = 64 -- the size bench.py --config stress times."""
    from chromoformer_amd import _lib
    Bq, S, H, L = 128, 16, 2, 800
    N = Bq * (S + 1)
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(0)
    proj = torch.randn(N, L, 3 * H * 64, device=dev, generator=g)
    q, k, v = proj[:, :, :128], proj[:, :, 128:256], proj[:, :, 256:]
    d_o = torch.randn(N, L, H * 64, device=dev, generator=g)
    # ragged key lengths as the dataset produces them (centred valid range), a few dummy (fully masked) regions
    gc = torch.Generator().manual_seed(1)
    n_valid = torch.randint(1, L + 1, (N,), generator=gc)
    n_valid[::97] = 0
    lo = (L - n_valid + 1) // 2
    ar = torch.arange(L)[None]
    kvalid = ((ar >= lo[:, None]) & (ar < (lo + n_valid)[:, None])).to(torch.uint8).to(dev)
    lib = _lib.lib()
    sh = _lib.cf_attn_shape(N, H, L, L, 3 * H * 64, 3 * H * 64, 3 * H * 64, H * 64)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda t: C.c_void_p(t.data_ptr())

    def run():
        o = torch.empty(N, L, H * 64, device=dev)
        stats = torch.empty(N, H, L, 2, device=dev)
        dproj = torch.zeros_like(proj)
        ws = torch.empty(N * H * L, device=dev)
        _lib.check(lib.cf_op_attention_fwd(C.byref(sh), p(q), p(k), p(v), None, p(kvalid), None, p(o), p(stats), st), "cf_op_attention_fwd")
        _lib.check(lib.cf_op_attention_bwd(C.byref(sh), p(q), p(k), p(v), None, p(kvalid), None, p(o), p(stats), p(d_o), p(dproj[:, :, :128]),
                                           p(dproj[:, :, 128:256]), p(dproj[:, :, 256:]), p(ws), st), "cf_op_attention_bwd")
        torch.cuda.synchronize()
        return o, dproj

    o1, d1 = run()
    o2, d2 = run()
    assert torch.equal(o1, o2) and torch.equal(d1, d2)                     # no atomics: bit-reproducible at full size
    assert torch.isfinite(o1).all() and torch.isfinite(d1).all()
    for n in (0, 97, 1234, N - 1):                                         # 97: a fully masked (dummy) region -> uniform softmax
        qq, kk, vv = (t[n:n + 1].detach().cpu().double().requires_grad_(True) for t in (q, k, v))
        mask4 = ~(kvalid[n:n + 1].cpu().bool()[:, None, None, :]).expand(1, 1, L, L)
        ctx = orc._attend(orc._split_heads(qq, H), orc._split_heads(kk, H), orc._split_heads(vv, H), mask4)
        ref = orc._merge_heads(ctx)
        ref.backward(d_o[n:n + 1].cpu().double())
        assert (o1[n:n + 1].cpu() - ref.detach().float()).abs().max() < 2e-5, n
        for got, r, name in ((d1[n:n + 1, :, :128], qq.grad, "dq"), (d1[n:n + 1, :, 128:256], kk.grad, "dk"), (d1[n:n + 1, :, 256:], vv.grad, "dv")):
            assert (got.cpu() - r.float()).abs().max() <= 1e-4 * max(r.abs().max().item(), 1e-3), (n, name)
