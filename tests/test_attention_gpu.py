"""Dense attention core (cf_op_attention_fwd / _bwd) against the oracle's `_attend` (modules.py:58-77 restated in
oracle/chromoformer_oracle.py) and its autograd: ragged lengths, valid-vector and arbitrary masks, fully masked
rows (uniform softmax, no NaN), cross-attention with Lq != Lk, the fused-projection layout (q|k|v chunks passed in
place), and run-to-run determinism.  Tolerance: 2e-5 absolute on outputs of O(1) magnitude (fp32, different
summation order), gradients 1e-4 of each tensor's max."""
import ctypes as C

import pytest
import torch

from oracle import chromoformer_oracle as orc

pytestmark = pytest.mark.gpu

@pytest.fixture(params=["split", "fused", "fused64"], autouse=True)
def attention_backward_form(request, monkeypatch):
    """Every test of this module runs three times: with the backward as two kernels split by output owner (k_attn_bwd_kv + k_attn_bwd_q), as the
    one-pass kernel with 128 keys per pass (k_attn_bwd2, round 6; the library picks a one-pass kernel by itself only for launches with >= 512
    (sequence, head) pairs) and as the one-pass kernel of rounds 4-5 (k_attn_bwd, 64 keys per pass: CF_ATTN_BWD_V1=1)."""
    monkeypatch.setenv("CF_ATTN_BWD_SPLIT", "1" if request.param == "split" else "-1")
    monkeypatch.setenv("CF_ATTN_BWD_V1", "1" if request.param == "fused64" else "0")
    return request.param



def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _run(q, k, v, H, qvalid=None, kvalid=None, mask=None, d_o=None):
    """q, k, v: cuda views [N, L, H*64] (any row stride).  Returns o (and dq, dk, dv)."""
    from chromoformer_amd import _lib
    L = _lib.lib()
    N, Lq, Lk = q.shape[0], q.shape[1], k.shape[1]
    sh = _lib.cf_attn_shape(N, H, Lq, Lk, q.stride(1), k.stride(1), v.stride(1), H * 64)
    o = torch.empty(N, Lq, H * 64, device=q.device)
    stats = torch.empty(N, H, Lq, 2, device=q.device)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.cf_op_attention_fwd(C.byref(sh), _ptr(q), _ptr(k), _ptr(v), _ptr(qvalid), _ptr(kvalid), _ptr(mask), _ptr(o), _ptr(stats), st),
               "cf_op_attention_fwd")
    if d_o is None:
        return o
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    assert dq.stride(1) == q.stride(1) or True
    dqc, dkc, dvc = (torch.zeros(t.shape, device=q.device) for t in (q, k, v))
    sh2 = _lib.cf_attn_shape(N, H, Lq, Lk, q.stride(1), k.stride(1), v.stride(1), H * 64)
    # gradients are written in the layouts of q, k, v: give them the same strides
    dq = torch.empty_strided(q.shape, q.stride(), device=q.device).zero_()
    dk = torch.empty_strided(k.shape, k.stride(), device=q.device).zero_()
    dv = torch.empty_strided(v.shape, v.stride(), device=q.device).zero_()
    ws = torch.empty(N * H * Lq, device=q.device)
    _lib.check(L.cf_op_attention_bwd(C.byref(sh2), _ptr(q), _ptr(k), _ptr(v), _ptr(qvalid), _ptr(kvalid), _ptr(mask), _ptr(o), _ptr(stats),
                                     _ptr(d_o.contiguous()), _ptr(dq), _ptr(dk), _ptr(dv), _ptr(ws), st), "cf_op_attention_bwd")
    return o, dq, dk, dv


def _oracle(q, k, v, H, mask4, d_o=None):
    q, k, v = (t.detach().cpu().double().requires_grad_(True) for t in (q, k, v))
    ctx = orc._attend(orc._split_heads(q, H), orc._split_heads(k, H), orc._split_heads(v, H), mask4)
    o = orc._merge_heads(ctx)
    if d_o is None:
        return o.float()
    o.backward(d_o.cpu().double())
    return o.float(), q.grad.float(), k.grad.float(), v.grad.float()


def _valid(N, L, gen, full_rows=()):
    v = torch.zeros(N, L, dtype=torch.uint8)
    for n in range(N):
        n_valid = int(torch.randint(1, L + 1, (1,), generator=gen))
        lo = (L - n_valid + 1) // 2
        v[n, lo:lo + n_valid] = 1
    for n in full_rows:
        v[n] = 0                                         # a dummy region: everything masked (data.py:196-198)
    return v


@pytest.mark.parametrize("N,H,Lq,Lk", [(3, 2, 80, 80), (2, 2, 400, 400), (2, 1, 37, 150), (1, 2, 800, 800), (5, 3, 64, 129)])
def test_forward_backward_match_oracle(N, H, Lq, Lk):
    gen = torch.Generator().manual_seed(N * 1000 + Lq + Lk)
    q, k, v = (torch.randn(N, L, H * 64, generator=gen) for L in (Lq, Lk, Lk))
    d_o = torch.randn(N, Lq, H * 64, generator=gen)
    qv, kv = _valid(N, Lq, gen), _valid(N, Lk, gen, full_rows=(N - 1,) if N > 2 else ())
    mask4 = ~(qv.bool()[:, None, :, None] & kv.bool()[:, None, None, :])
    ref = _oracle(q, k, v, H, mask4, d_o)
    got = _run(q.cuda(), k.cuda(), v.cuda(), H, qv.cuda(), kv.cuda(), None, d_o.cuda())
    assert torch.isfinite(got[0]).all()
    assert (got[0].cpu() - ref[0]).abs().max() < 2e-5
    for g, r, name in zip(got[1:], ref[1:], ("dq", "dk", "dv")):
        assert (g.cpu() - r).abs().max() <= 1e-4 * max(r.abs().max().item(), 1e-3), name


def test_arbitrary_mask_no_mask_and_fused_projection_layout():
    gen = torch.Generator().manual_seed(5)
    N, H, L = 2, 2, 96
    proj = torch.randn(N, L, 3 * H * 64, generator=gen).cuda()           # att(x): q | k | v chunks, head-major inside
    q, k, v = proj[:, :, :128], proj[:, :, 128:256], proj[:, :, 256:]
    d_o = torch.randn(N, L, H * 64, generator=gen).cuda()
    mask = (torch.rand(N, L, L, generator=gen) < 0.3).to(torch.uint8)
    mask[0, 7] = 1                                                        # one fully masked row
    for m in (mask, None):
        mask4 = m.bool()[:, None] if m is not None else None
        ref = _oracle(q, k, v, H, mask4, d_o)
        got = _run(q, k, v, H, None, None, m.cuda() if m is not None else None, d_o)
        assert (got[0].cpu() - ref[0]).abs().max() < 2e-5
        for g, r in zip(got[1:], ref[1:]):
            assert (g.cpu() - r).abs().max() <= 1e-4 * r.abs().max().item()
    # the fully masked row is the mean of all values (uniform softmax), exactly like the reference
    o = _run(q, k, v, H, None, None, mask.cuda())
    assert (o[0, 7].cpu() - v[0].mean(0).cpu()).abs().max() < 2e-6


def test_bit_reproducible():
    gen = torch.Generator().manual_seed(9)
    q, k, v, d_o = (torch.randn(4, 200, 128, generator=gen).cuda() for _ in range(4))
    a = _run(q, k, v, 2, None, None, None, d_o)
    b = _run(q, k, v, 2, None, None, None, d_o)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("N,H,Lq,Lk,masked", [(3, 2, 800, 800, False), (2, 1, 37, 150, False), (4, 2, 130, 65, False), (2, 2, 96, 200, True)])
def test_transposed_forward_equals_the_round_1_kernel(N, H, Lq, Lk, masked, monkeypatch):
    """k_attn_fwd (round 5: transposed score tiles, P in registers, the mask as one fused multiply-add per score, 128 query rows per workgroup,
    key blocks beyond the tensor skipped) against k_attn_fwd_v1 (CF_ATTN_FWD_V1=1: 64 rows per workgroup, P through LDS, per-score mask test):
    same outputs and the same row statistics (maximum bit for bit; 1 / sum to fp32 rounding -- the sums add the same terms in another order) --
    padded keys, padded and fully masked query rows, lengths that are no multiple of any tile, with and without a byte mask."""
    from chromoformer_amd import _lib
    gen = torch.Generator().manual_seed(Lq * 7 + Lk)
    q, k, v = (torch.randn(N, L, H * 64, generator=gen).cuda() for L in (Lq, Lk, Lk))
    qv, kv = _valid(N, Lq, gen).cuda(), _valid(N, Lk, gen, full_rows=(N - 1,)).cuda()
    mask = (torch.rand(N, Lq, Lk, generator=gen) < 0.4).to(torch.uint8)
    mask[0, 3] = 1
    mask = mask.cuda() if masked else None
    out = {}
    for form in ("0", "1"):
        monkeypatch.setenv("CF_ATTN_FWD_V1", form)
        L = _lib.lib()
        sh = _lib.cf_attn_shape(N, H, Lq, Lk, q.stride(1), k.stride(1), v.stride(1), H * 64)
        o = torch.full((N, Lq, H * 64), float("nan"), device=q.device)
        stats = torch.full((N, H, Lq, 2), float("nan"), device=q.device)
        _lib.check(L.cf_op_attention_fwd(C.byref(sh), _ptr(q), _ptr(k), _ptr(v), _ptr(qv), _ptr(kv), _ptr(mask), _ptr(o), _ptr(stats),
                                         torch.cuda.current_stream().cuda_stream), "cf_op_attention_fwd")
        torch.cuda.synchronize()
        out[form] = (o, stats)
    (o0, s0), (o1, s1) = out["0"], out["1"]
    assert torch.isfinite(o0).all() and torch.isfinite(s0).all()
    assert (o0 - o1).abs().max() < 2e-6
    assert torch.equal(s0[..., 0], s1[..., 0])
    assert ((s0[..., 1] - s1[..., 1]).abs() <= 2e-6 * s1[..., 1].abs()).all()


def test_stress_length_properties_of_the_forward():
    """Size-independent properties at the stress configuration's length (L = 800: 6.25 workgroups of 128 query rows, 12.5 key tiles -- the ragged
    ends of both tilings), where the oracle takes too long for more than a handful of sequences: every output row is a convex combination of the
    value rows -- V = 1 gives 1, a V that is constant per key block of 16 gives a row inside [min, max] --, the output is linear in V, the saved
    statistics reproduce the probabilities (sum_j exp(s_ij - m_i) * inv_i = 1 for a sampled set of rows), and padded keys carry no weight."""
    from chromoformer_amd import _lib
    gen = torch.Generator().manual_seed(800)
    N, H, L = 6, 2, 800
    q, k = (torch.randn(N, L, H * 64, generator=gen).cuda() for _ in range(2))
    v1, v2 = (torch.randn(N, L, H * 64, generator=gen).cuda() for _ in range(2))
    kv = _valid(N, L, gen).cuda()
    qv = torch.ones(N, L, dtype=torch.uint8).cuda()

    def fwd(v):
        sh = _lib.cf_attn_shape(N, H, L, L, q.stride(1), k.stride(1), v.stride(1), H * 64)
        o = torch.empty(N, L, H * 64, device=q.device)
        stats = torch.empty(N, H, L, 2, device=q.device)
        _lib.check(_lib.lib().cf_op_attention_fwd(C.byref(sh), _ptr(q), _ptr(k), _ptr(v), _ptr(qv), _ptr(kv), None, _ptr(o), _ptr(stats),
                                                  torch.cuda.current_stream().cuda_stream), "cf_op_attention_fwd")
        return o, stats

    ones, stats = fwd(torch.ones_like(v1))
    assert (ones - 1).abs().max() < 2e-6
    o1, o2, o12 = fwd(v1)[0], fwd(v2)[0], fwd(v1 + 2 * v2)[0]
    assert (o12 - (o1 + 2 * o2)).abs().max() < 2e-5
    # padded keys carry no weight: values at padded keys do not matter
    v3 = v1.clone()
    v3[(kv == 0)[:, :, None].expand_as(v3)] = 1e6
    assert torch.equal(fwd(v3)[0], o1)
    # the saved statistics are the softmax's: recompute a few rows on the host
    for n, h, i in ((0, 0, 0), (1, 1, 799), (3, 0, 417), (5, 1, 128)):
        s = (q[n, i, h * 64:(h + 1) * 64].double() @ k[n, :, h * 64:(h + 1) * 64].double().T) / 8.0
        s[kv[n] == 0] = -1e9
        p = torch.softmax(s, 0)
        m, inv = stats[n, h, i].double()
        assert abs(float(m) - float(s.max())) < 1e-4
        assert abs(float((torch.exp(s - m) * inv).sum()) - 1) < 1e-5
        assert (p.float() @ v1[n, :, h * 64:(h + 1) * 64] - o1[n, i, h * 64:(h + 1) * 64]).abs().max() < 2e-5
