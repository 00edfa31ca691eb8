"""Stage-by-stage CPU statement of the RESTRUCTURED algorithm the HIP kernels execute
--  TEST INFRASTRUCTURE, NOT PRODUCT CODE  (same import rules as the rest of oracle/).

The dense oracle (chromoformer_oracle.py) restates what the reference executes.  The
HIP path computes the same function with two exact algebraic restructurings
(DESIGN.md section 2):

  1. centre query row only: the Embedding / Pairwise outputs are consumed at row L//2
     only (net.py:59, :138) and the Pairwise key/value stream never depends on the query
     stream (modules.py:159-160, :219), so one query row per sequence is evaluated;
  2. weight absorption for a single query: with one query q per sequence,
        score_j = q . (Wk x_j)        = (Wk^T q) . x_j
        ctx     = sum_j p_j (Wv x_j)  = Wv (sum_j p_j x_j)
     and with x_j = Wlp f_j + PE_j (f_j the 7 histone marks of bin j)
        score_j = f_j . (Wlp^T qt) + PE_j . qt,      qt = Wk^T q
        xbar    = Wlp (sum_j p_j f_j) + sum_j p_j PE_j
     so the per-bin work is two length-(7+128) dot products per head instead of a
     128x256 projection of every bin.

This module spells those stages out in torch (autograd supplies the backward the
hand-written kernels are checked against) and names every intermediate the kernels
expose through cf_debug_buffer().  tests/test_restructured.py proves it equal to the
dense oracle (forward 1e-6, gradients 1e-5 relative) on CPU.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .chromoformer_oracle import _cfg, positional_table


def centre_attention(qt, feats, pe, w_lp, mask_row, dh, keep=None, tag=""):
    """qt [N,H,D], feats [N,L,F], pe [L,D], w_lp [D,F], mask_row [N,L] bool -> xbar [N,H,D]."""
    u = torch.matmul(qt, w_lp)                                        # [N,H,F]
    s = (torch.einsum("nlf,nhf->nhl", feats, u) + torch.einsum("le,nhe->nhl", pe, qt)) / (dh ** 0.5)
    s = s.masked_fill(mask_row[:, None, :], -1e9)
    p = F.softmax(s, dim=-1)
    w = torch.einsum("nhl,nlf->nhf", p, feats)
    pi = torch.einsum("nhl,le->nhe", p, pe)
    xbar = torch.matmul(w, w_lp.t()) + pi
    if keep is not None:
        keep[tag + "p"] = p
        keep[tag + "w"] = w
    return xbar


def _post_chain(P, att_pre, ff_pre, x, a, keep, tag):
    """y1 = LN(x + a Wo^T + bo); out = LN(y1 + relu(y1 W1^T + b1) W2^T + b2)."""
    t1 = x + F.linear(a, P[att_pre + "ff.weight"], P[att_pre + "ff.bias"])
    y1 = F.layer_norm(t1, t1.shape[-1:], P[att_pre + "ln.weight"], P[att_pre + "ln.bias"])
    h = F.relu(F.linear(y1, P[ff_pre + "l1.weight"], P[ff_pre + "l1.bias"]))
    t2 = y1 + F.linear(h, P[ff_pre + "l2.weight"], P[ff_pre + "l2.bias"])
    out = F.layer_norm(t2, t2.shape[-1:], P[ff_pre + "ln.weight"], P[ff_pre + "ln.bias"])
    keep[tag + "y1"] = y1
    keep[tag + "hdn"] = h
    keep[tag + "out"] = out
    return out


def _centre_layer(P, att_pre, ff_pre, wq, wk, wv, x, feats, pe, w_lp, mask_row, n_heads, keep, tag):
    """One centre-row attention layer (Embedding layer or Pairwise layer)."""
    N, D = x.shape
    dm = wq.shape[0]
    dh = dm // n_heads
    q = F.linear(x, wq)                                               # [N, dm]
    qt = torch.einsum("nhd,hde->nhe", q.view(N, n_heads, dh), wk.view(n_heads, dh, D))
    keep[tag + "q"], keep[tag + "qt"] = q, qt
    xbar = centre_attention(qt, feats, pe, w_lp, mask_row, dh, keep, tag)
    keep[tag + "xbar"] = xbar
    a = torch.einsum("nhe,hde->nhd", xbar, wv.view(n_heads, dh, D)).reshape(N, dm)
    keep[tag + "a"] = a
    return _post_chain(P, att_pre, ff_pre, x, a, keep, tag)


def forward(P, batch, cfg=None, keep=None):
    """Same inputs / output as chromoformer_oracle.forward; centre-row restructured."""
    c = _cfg(cfg)
    keep = {} if keep is None else keep
    D = c["d_emb"]
    e, p, r = c["embed"], c["pairwise_interaction"], c["regulation"]
    assert e["n_layers"] == 1, "centre-row evaluation needs a single Embedding layer"
    heads_in, heads_out = [], []
    for b in c["binsizes"]:
        pf, pm = batch["promoter_feats"][b], batch["promoter_pad_masks"][b]
        cf, cm = batch["pcre_feats"][b], batch["pcre_pad_masks"][b]
        B, S, L = cf.shape[0], cf.shape[1], cf.shape[2]
        ctr = L // 2
        pe = positional_table(L, D)
        # ---- Embedding, centre row ------------------------------------------------
        pre = "embed.%d." % b
        lp = pre + "transformer.layers.0."
        w_lp = P[pre + "lin_proj.weight"]
        feats = pf[:, 0]                                              # [B,L,F]
        x0 = F.linear(feats[:, ctr], w_lp) + pe[ctr]
        keep["E%d.x0" % b] = x0
        w_att = P[lp + "self_att.att.weight"]
        dm = e["d_model"]
        e_c = _centre_layer(P, lp + "self_att.", lp + "ff.", w_att[:dm], w_att[dm:2 * dm], w_att[2 * dm:3 * dm],
                            x0, feats, pe, w_lp, pm[:, 0, 0, ctr, :], e["n_heads"], keep, "E%d." % b)
        # ---- Pairwise, centre row -------------------------------------------------
        pre = "pairwise_interaction.%d." % b
        w_lpc = P[pre + "lin_proj_pcre.weight"]
        xp = F.linear(e_c, P[pre + "lin_proj_p.weight"])             # [B,D]
        keep["P%d.xp0" % b] = xp
        xp = xp[:, None, :].expand(B, S, D).reshape(B * S, D)
        feats_c = cf.reshape(B * S, L, -1)
        mrow = cm[:, :, 0, ctr, :].reshape(B * S, L)
        dmp = p["d_model"]
        for l in range(p["n_layers"]):
            lp = pre + "transformer.layers.%d." % l
            w_c = P[lp + "self_att.c_att.weight"]
            xp = _centre_layer(P, lp + "self_att.", lp + "ff.", P[lp + "self_att.p_att.weight"], w_c[:dmp], w_c[dmp:],
                               xp, feats_c, pe, w_lpc, mrow, p["n_heads"], keep, "P%d.%d." % (b, l))
        z = xp.view(B, S, D)
        # ---- Regulation -----------------------------------------------------------
        x = torch.cat([e_c[:, None, :], z], dim=1)                    # [B,T,D]
        keep["R%d.x0" % b] = x
        heads_in.append(x[:, 0])
        pre = "regulation.%d." % b
        T = S + 1
        H, dmr = r["n_heads"], r["d_model"]
        dh = dmr // H
        mask = batch["interaction_masks"][b][:, 0]                    # [B,T,T]
        freq = batch["interaction_freq"]
        for l in range(r["n_layers"]):
            lp = pre + "transformer.layers.%d." % l
            tag = "R%d.%d." % (b, l)
            qkvg = F.linear(x, P[lp + "self_att.att.weight"])         # [B,T,4*dm]
            keep[tag + "qkvg"] = qkvg
            q, k, v, g = (t.view(B, T, H, dh).transpose(1, 2) for t in qkvg.chunk(4, dim=-1))
            s = torch.matmul(q, k.transpose(-1, -2)) / (dh ** 0.5)
            s = s + P[lp + "self_att.gamma_f"].view(1, H, 1, 1) * freq[:, None]
            s = s.masked_fill(mask[:, None], -1e9)
            pr = F.softmax(s, dim=-1)
            o = torch.matmul(pr, v) * torch.sigmoid(g)
            a = o.transpose(1, 2).reshape(B, T, dmr)
            keep[tag + "p"], keep[tag + "a"] = pr, a
            x = _post_chain(P, lp + "self_att.", lp + "ff.", x.reshape(B * T, D), a.reshape(B * T, dmr), keep, tag)
            x = x.view(B, T, D)
        heads_out.append(x[:, 0])
    h_in = torch.cat(heads_out, dim=1) + torch.cat(heads_in, dim=1)
    keep["H.in"] = h_in
    h1 = F.relu(F.linear(h_in, P["fc_head.0.weight"], P["fc_head.0.bias"]))
    keep["H.h1"] = h1
    return F.linear(h1, P["fc_head.2.weight"], P["fc_head.2.bias"])
