"""The all-rows Embedding path (csrc/cf_embed_full.h + the dense transformer layer): embed.n_layers > 1 through the model
classes, and EmbeddingTransformer's full output (net.py:9-59, modules.py:104-124), against the oracle and its autograd."""
import numpy as np
import pytest
import torch

from oracle import chromoformer_oracle as orc

pytestmark = pytest.mark.gpu
CFG2 = {"embed": {"n_layers": 2, "n_heads": 2, "d_model": 128, "d_ff": 128}}


def _perturb(P, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for v in P.values():
            v.add_(0.05 * torch.randn(v.shape, generator=g))
    return P


def _call(model, batch):
    return model(batch["promoter_feats"], batch["promoter_pad_masks"], batch["pcre_feats"], batch["pcre_pad_masks"],
                 batch["interaction_masks"], batch["interaction_freq"])


@pytest.mark.parametrize("reg", [False, True])
def test_two_embedding_layers_forward_backward_adamw(reg):
    from chromoformer_amd import ChromoformerClassifier, ChromoformerRegressor
    Model = ChromoformerRegressor if reg else ChromoformerClassifier
    model = Model(embed_kws=CFG2["embed"], seed=42, max_batch=5).cuda(0)
    P = _perturb(orc.init_params(CFG2, 42, reg), 3)
    assert list(model.state_dict()) == list(P) and len(P) == 370 + 3 * 13       # 13 more tensors per resolution
    model.load_state_dict(P)
    for t in P.values():
        t.requires_grad_(True)
    batch = orc.synthetic_batch(5, seed=21, regime="realistic", regression=reg)
    with torch.no_grad():
        ref = orc.forward(P, batch, CFG2)
        out = _call(model, batch).cpu()
    assert (out - ref).abs().max() < 1e-4
    # one full optimisation step: gradients of every tensor (both Embedding layers included), then AdamW
    opt = orc.make_optimizer(P, 3e-5)
    loss_ref, _ = orc.train_step(P, opt, batch, CFG2, regression=reg)
    grads = {k: v.grad.clone() for k, v in P.items() if v.grad is not None}
    packed = model.pack_batch(batch)
    logits, loss = model.forward_backward(packed, batch["label"])
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 1e-4
    model._publish_grads()
    named = dict(model.named_parameters())
    for k, gref in grads.items():
        got = named[k].grad.cpu()
        assert (got - gref).abs().max() <= 1e-3 * max(gref.abs().max().item(), 1e-6), k
    assert any(k.startswith("embed.100.transformer.layers.1.") for k in grads)
    model.adamw_step(3e-5)
    torch.cuda.synchronize()
    sd = model.state_dict()
    # first AdamW update u = lr g / (|g| + eps): insensitive to rounding noise in g where |g| >> eps = 1e-8, amplified up to a
    # fraction of lr = 3e-5 where |g| ~ eps (same treatment as tests/test_gpu_parity.py)
    for k in P:
        d = (sd[k].cpu() - P[k].detach()).abs()
        if P[k].grad is None:
            assert d.max().item() == 0.0, k
            continue
        big = P[k].grad.abs() > 1e-6
        assert (d[big].max().item() < 5e-7) if big.any() else True, k
        assert d.max().item() <= 3e-5, k


def test_trainer_replays_the_all_rows_path_from_a_graph():
    """The dense-layer operators build their tile tables on the device from by-value arguments: nothing synchronises, so the
    step with two Embedding layers is captured and replayed like the default one -- bit-identical to eager launches."""
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    res = []
    for graph in (False, True):
        model = ChromoformerClassifier(embed_kws=CFG2["embed"], seed=42, max_batch=4).cuda(0)
        tr = Trainer(model, lr=3e-5, use_graph=graph)
        slot = tr.stage(orc.synthetic_batch(4, seed=2, regime="dense"))
        losses = []
        for _ in range(3):
            _, loss = tr.step(slot)
            tr.stream.synchronize()
            losses.append(float(loss))
        assert losses[2] < losses[0] and all(np.isfinite(losses))
        res.append((losses, {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        assert torch.equal(res[0][1][k], res[1][1][k]), k


@pytest.mark.parametrize("compact", [False, True])
def test_full_promoter_embedding_of_the_default_model(compact):
    """EmbeddingTransformer.forward()[0] for the default single-layer model: every row, not just the centre one."""
    from chromoformer_amd import ChromoformerClassifier
    model = ChromoformerClassifier(seed=42, max_batch=6).cuda(0)
    P = _perturb(orc.init_params(None, 42, False), 5)
    model.load_state_dict(P)
    batch = orc.synthetic_batch(6, seed=8, regime="realistic")
    # a promoter with padding on both sides (w_prom < w_max): rows and columns outside [lo, hi) masked
    for b, m in batch["promoter_pad_masks"].items():
        L = m.shape[-1]
        v = torch.ones(L, dtype=torch.bool)
        v[: L // 5] = False
        v[L - L // 7:] = False
        m[1, 0, 0] = ~(v[:, None] & v[None, :])
    masks = batch["promoter_pad_masks"]
    if compact:
        masks = {b: m[:, 0, 0, m.shape[-1] // 2, :].contiguous() for b, m in masks.items()}
    got = model.embed_full(batch["promoter_feats"], masks)
    cfg = orc._cfg(None)
    with torch.no_grad():
        for b in (2000, 500, 100):
            full, centre = orc.embed_forward(P, "embed.%d." % b, cfg, batch["promoter_feats"][b], batch["promoter_pad_masks"][b])
            assert got[b].shape == full.shape
            assert (got[b].cpu() - full).abs().max() < 2e-5, b


def test_arbitrary_bool_masks_through_two_embedding_layers():
    """modules.py:71-73 takes ANY bool mask (the reference's own smoke block feeds `randn(...).bool()`-style ones, net.py:431-568).  With
    embed.n_layers = 2 every row of the promoter mask is read: model(...) hands the library the full [B, 1, L, L] tensor and matches the
    oracle (forward 1e-4, gradients 1e-3); a Slot, which keeps only the centre row, refuses such a mask by name instead of replacing it."""
    from chromoformer_amd import ChromoformerClassifier
    from chromoformer_amd.engine import Trainer
    model = ChromoformerClassifier(embed_kws=CFG2["embed"], seed=42, max_batch=4).cuda(0)
    P = _perturb(orc.init_params(CFG2, 42, False), 5)
    model.load_state_dict(P)
    batch = orc.synthetic_batch(4, seed=33, regime="realistic")
    g = torch.Generator().manual_seed(9)
    for b in batch["promoter_pad_masks"]:      # unstructured masks, ~30 % of the entries set, one query row fully masked
        m = torch.rand(batch["promoter_pad_masks"][b].shape, generator=g) < 0.3
        m[1, 0, 0, 3, :] = True
        batch["promoter_pad_masks"][b] = m
        c = torch.rand(batch["pcre_pad_masks"][b].shape, generator=g) < 0.3
        batch["pcre_pad_masks"][b] = c
    for t in P.values():
        t.requires_grad_(True)
    ref = orc.forward(P, batch, CFG2)
    loss_ref = orc.loss_fn(ref, batch["label"], False)
    loss_ref.backward()
    with torch.no_grad():
        out = _call(model, batch).cpu()
    assert (out - ref.detach()).abs().max() < 1e-4
    logits, loss = model.forward_backward(model.pack_batch(batch), batch["label"])
    torch.cuda.synchronize()
    assert abs(float(loss) - float(loss_ref)) < 1e-4
    model._publish_grads()
    named = dict(model.named_parameters())
    n = 0
    for k, v in P.items():
        if v.grad is None:
            continue
        got = named[k].grad.cpu()
        assert (got - v.grad).abs().max() <= 1e-3 * max(v.grad.abs().max().item(), 1e-6), k
        n += 1
    assert n > 300
    with pytest.raises(ValueError, match="not of the dataset's form"):
        Trainer(model, lr=3e-5).stage(batch)
    # ... while the dataset's structured masks go through a slot as before
    ok = orc.synthetic_batch(4, seed=33, regime="realistic")
    Trainer(model, lr=3e-5).stage(ok)
