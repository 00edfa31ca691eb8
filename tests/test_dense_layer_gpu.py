"""cf_op_dense_layer_fwd: one AttentionBlock / PairwiseAttentionBlock over ALL rows against the oracle's
self_attention_block / pairwise_attention_block + feed_forward (modules.py:104-112, 198-208 restated), seeded
reference-initialised weights, ragged lengths, a dummy (fully masked) pCRE.  Tolerance 2e-5 on O(1) outputs."""
import ctypes as C

import pytest
import torch

from oracle import chromoformer_oracle as orc

pytestmark = pytest.mark.gpu

@pytest.fixture(params=["split", "fused"], autouse=True)
def attention_backward_form(request, monkeypatch):
    """Every test of this module runs twice: with the backward as two kernels split by output owner (k_attn_bwd_kv + k_attn_bwd_q)
    and as the one-pass kernel (k_attn_bwd; the library picks it by itself only for launches with >= 512 (sequence, head) pairs)."""
    monkeypatch.setenv("CF_ATTN_BWD_SPLIT", "1" if request.param == "split" else "-1")
    return request.param



def _layer(P, pre, wq, wkv, dev):
    from chromoformer_amd import _lib
    keep = {k: P[pre + k].to(dev).contiguous() for k in ("self_att.ff.weight", "self_att.ff.bias", "self_att.ln.weight", "self_att.ln.bias",
                                                       "ff.l1.weight", "ff.l1.bias", "ff.l2.weight", "ff.l2.bias", "ff.ln.weight", "ff.ln.bias")}
    keep["wq"], keep["wkv"] = wq.to(dev).contiguous(), wkv.to(dev).contiguous()
    w = _lib.cf_dense_layer()
    for f, k in (("wq", "wq"), ("wkv", "wkv"), ("wo", "self_att.ff.weight"), ("bo", "self_att.ff.bias"), ("ln1_g", "self_att.ln.weight"),
                 ("ln1_b", "self_att.ln.bias"), ("w1", "ff.l1.weight"), ("b1", "ff.l1.bias"), ("w2", "ff.l2.weight"), ("b2", "ff.l2.bias"),
                 ("ln2_g", "ff.ln.weight"), ("ln2_b", "ff.ln.bias")):
        setattr(w, f, keep[k].data_ptr())
    w.d_ff = keep["ff.l1.weight"].shape[0]
    return w, keep


def _run(w, x_q, x_kv, qv, kv, N, Lq, Lk):
    from chromoformer_amd import _lib
    L = _lib.lib()
    ws = torch.empty(L.cf_op_dense_layer_workspace(N, Lq, Lk, w.d_ff), device=x_q.device)
    y = torch.empty(N, Lq, 128, device=x_q.device)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    _lib.check(L.cf_op_dense_layer_fwd(C.byref(w), p(x_q), p(x_kv), p(qv), p(kv), None, N, Lq, Lk, p(y), p(ws),
                                       torch.cuda.current_stream().cuda_stream), "cf_op_dense_layer_fwd")
    return y


@pytest.mark.parametrize("res,L", [(2000, 20), (500, 80), (100, 400), (100, 800)])      # 800: the stress configuration's sequence length
def test_embedding_layer_all_rows(res, L):
    dev = torch.device("cuda", 0)
    P = orc.init_params(None, 5, False)
    pre = "embed.%d.transformer.layers.0." % res
    att = P[pre + "self_att.att.weight"]
    w, keep = _layer(P, pre, att[:128], att[128:], dev)
    g = torch.Generator().manual_seed(L)
    N = 3
    x = torch.randn(N, L, 128, generator=g)
    valid = torch.zeros(N, L, dtype=torch.uint8)
    for n, nv in enumerate((L, max(1, L // 3), L - 1)):
        lo = (L - nv + 1) // 2
        valid[n, lo:lo + nv] = 1
    mask4 = ~(valid.bool()[:, None, :, None] & valid.bool()[:, None, None, :])
    with torch.no_grad():
        ref = orc.feed_forward(P, pre + "ff.", orc.self_attention_block(P, pre + "self_att.", x, mask4, None, 2, False))
    xd, vd = x.to(dev), valid.to(dev)
    got = _run(w, xd, xd, vd, vd, N, L, L).cpu()
    assert (got - ref).abs().max() < 2e-5


def test_pairwise_layer_all_rows():
    dev = torch.device("cuda", 0)
    P = orc.init_params(None, 6, False)
    pre = "pairwise_interaction.500.transformer.layers.1."
    w, keep = _layer(P, pre, P[pre + "self_att.p_att.weight"], P[pre + "self_att.c_att.weight"], dev)
    assert w.d_ff == 256
    g = torch.Generator().manual_seed(1)
    N, L = 4, 80
    x_p, x_c = torch.randn(N, L, 128, generator=g), torch.randn(N, L, 128, generator=g)
    pv = torch.ones(N, L, dtype=torch.uint8)
    cv = torch.zeros(N, L, dtype=torch.uint8)
    cv[0, 30:50] = 1
    cv[1, :] = 1
    cv[2, 39:41] = 1                                   # cv[3] stays all zero: a dummy pCRE, uniform attention
    mask4 = ~(pv.bool()[:, None, :, None] & cv.bool()[:, None, None, :])
    with torch.no_grad():
        ref = orc.feed_forward(P, pre + "ff.", orc.pairwise_attention_block(P, pre + "self_att.", x_p, x_c, mask4, 2))
    got = _run(w, x_p.to(dev), x_c.to(dev), pv.to(dev), cv.to(dev), N, L, L).cpu()
    assert torch.isfinite(got).all() and (got - ref).abs().max() < 2e-5


def _train(w, keep, x_q, x_kv, qv, kv, dy, N, Lq, Lk):
    from chromoformer_amd import _lib
    L = _lib.lib()
    dev = x_q.device
    ws = torch.empty(L.cf_op_dense_layer_train_workspace(N, Lq, Lk, w.d_ff), device=dev)
    tables = torch.empty(1 << 20, device=dev)                       # 4 MiB
    y = torch.empty(N, Lq, 128, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.cf_op_dense_layer_fwd_train(C.byref(w), p(x_q), p(x_kv), p(qv), p(kv), None, N, Lq, Lk, p(y), p(ws), st), "fwd_train")
    names = ("wq", "wkv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")
    src = dict(wo="self_att.ff.weight", bo="self_att.ff.bias", ln1_g="self_att.ln.weight", ln1_b="self_att.ln.bias", w1="ff.l1.weight",
               b1="ff.l1.bias", w2="ff.l2.weight", b2="ff.l2.bias", ln2_g="ff.ln.weight", ln2_b="ff.ln.bias", wq="wq", wkv="wkv")
    grads = {n: torch.full_like(keep[src[n]], float("nan")) for n in names}
    g = _lib.cf_dense_layer_grads()
    for n in names:
        setattr(g, n, grads[n].data_ptr())
    dxq, dxk = torch.empty_like(x_q), torch.empty_like(x_kv)
    _lib.check(L.cf_op_dense_layer_bwd(C.byref(w), p(x_q), p(x_kv), p(qv), p(kv), None, N, Lq, Lk, p(dy), p(dxq), p(dxk), C.byref(g), p(ws),
                                       p(tables), st), "bwd")
    torch.cuda.synchronize()
    return y, dxq, dxk, grads


def _close(got, ref, name, tol=2e-4):
    err = (got.cpu() - ref).abs().max().item()
    assert err <= tol * max(ref.abs().max().item(), 1e-3), (name, err, ref.abs().max().item())


@pytest.mark.parametrize("L,N", [(80, 3), (400, 2), (800, 4), (96, 70), (96, 140)])      # (800, 4): the stress shape; the last ones span several split-K chunks (6,720 / 13,440 rows; the largest also two chunks of bias partials)
def test_embedding_layer_backward(L, N):
    dev = torch.device("cuda", 0)
    P = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(None, 5, False).items()}
    pre = "embed.500.transformer.layers.0."
    att = P[pre + "self_att.att.weight"]
    w, keep = _layer({k: v.detach() for k, v in P.items()}, pre, att.detach()[:128], att.detach()[128:], dev)
    g = torch.Generator().manual_seed(L + N)
    x = torch.randn(N, L, 128, generator=g).requires_grad_(True)
    dy = torch.randn(N, L, 128, generator=g)
    valid = torch.ones(N, L, dtype=torch.uint8)
    valid[0, : L // 4] = 0
    valid[N - 1, L // 2:] = 0
    mask4 = ~(valid.bool()[:, None, :, None] & valid.bool()[:, None, None, :])
    ref = orc.feed_forward(P, pre + "ff.", orc.self_attention_block(P, pre + "self_att.", x, mask4, None, 2, False))
    ref.backward(dy)
    xd, vd = x.detach().to(dev), valid.to(dev)
    y, dxq, dxk, grads = _train(w, keep, xd, xd, vd, vd, dy.to(dev), N, L, L)
    assert (y.cpu() - ref.detach()).abs().max() < 2e-5
    _close(dxq + dxk, x.grad, "dx")
    _close(torch.cat([grads["wq"], grads["wkv"]]), att.grad, "att.weight")
    for n, k in (("wo", "self_att.ff.weight"), ("bo", "self_att.ff.bias"), ("ln1_g", "self_att.ln.weight"), ("ln1_b", "self_att.ln.bias"),
                 ("w1", "ff.l1.weight"), ("b1", "ff.l1.bias"), ("w2", "ff.l2.weight"), ("b2", "ff.l2.bias"), ("ln2_g", "ff.ln.weight"),
                 ("ln2_b", "ff.ln.bias")):
        _close(grads[n], P[pre + k].grad, k)


@pytest.mark.parametrize("N,Lq,Lk", [(3, 64, 150), (4, 800, 800)])      # (4, 800, 800): the stress shape
def test_pairwise_layer_backward(N, Lq, Lk):
    dev = torch.device("cuda", 0)
    P = {k: v.clone().requires_grad_(True) for k, v in orc.init_params(None, 6, False).items()}
    pre = "pairwise_interaction.100.transformer.layers.0."
    Pd = {k: v.detach() for k, v in P.items()}
    w, keep = _layer(Pd, pre, Pd[pre + "self_att.p_att.weight"], Pd[pre + "self_att.c_att.weight"], dev)
    g = torch.Generator().manual_seed(3)
    x_p = torch.randn(N, Lq, 128, generator=g).requires_grad_(True)
    x_c = torch.randn(N, Lk, 128, generator=g).requires_grad_(True)
    dy = torch.randn(N, Lq, 128, generator=g)
    pv = torch.ones(N, Lq, dtype=torch.uint8)
    cv = torch.zeros(N, Lk, dtype=torch.uint8)
    cv[0, 40:90] = 1
    cv[1, :] = 1                                       # cv[2] all zero: dummy pCRE
    if N > 3:
        cv[3, Lk // 2 - 3: Lk // 2 + 4] = 1
    mask4 = ~(pv.bool()[:, None, :, None] & cv.bool()[:, None, None, :])
    ref = orc.feed_forward(P, pre + "ff.", orc.pairwise_attention_block(P, pre + "self_att.", x_p, x_c, mask4, 2))
    ref.backward(dy)
    y, dxq, dxk, grads = _train(w, keep, x_p.detach().to(dev), x_c.detach().to(dev), pv.to(dev), cv.to(dev), dy.to(dev), N, Lq, Lk)
    assert (y.cpu() - ref.detach()).abs().max() < 2e-5
    _close(dxq, x_p.grad, "dx_p")
    _close(dxk, x_c.grad, "dx_c")
    _close(grads["wq"], P[pre + "self_att.p_att.weight"].grad, "p_att")
    _close(grads["wkv"], P[pre + "self_att.c_att.weight"].grad, "c_att")
    for n, k in (("wo", "self_att.ff.weight"), ("w1", "ff.l1.weight"), ("w2", "ff.l2.weight"), ("b1", "ff.l1.bias"), ("ln2_g", "ff.ln.weight")):
        _close(grads[n], P[pre + k].grad, k)
