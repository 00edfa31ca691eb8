"""The HIP path away from the default configuration (everything INTEGRATION.md section C lists as supported):
other partner counts (run-time token count in the fused Regulation kernels; i_max = 16 is 17 tokens, one more than
an MFMA tile, and takes the unfused kernels), other depths and FFN widths, other window / bin sizes.  Forward and
all gradients against the oracle's autograd, same tolerances as the default-config tests."""
import pytest
import torch

from oracle import chromoformer_oracle as orc

pytestmark = pytest.mark.gpu
LOGIT_TOL, GRAD_TOL = 1e-4, 1e-3

VARIANTS = {
    "i_max4": dict(i_max=4),
    "i_max12": dict(i_max=12),
    "i_max16_unfused": dict(i_max=16),
    "shallow_narrow": dict(regulation=dict(n_layers=2, n_heads=8, d_model=256, d_ff=128),
                           pairwise_interaction=dict(n_layers=1, n_heads=2, d_model=128, d_ff=128),
                           embed=dict(n_layers=1, n_heads=2, d_model=128, d_ff=256)),
    "three_pairwise_layers": dict(pairwise_interaction=dict(n_layers=3, n_heads=2, d_model=128, d_ff=256)),
    "deep_reg": dict(regulation=dict(n_layers=8, n_heads=8, d_model=256, d_ff=256)),
    "other_bins": dict(binsizes=[1000, 250, 50], w_max=20000),          # L = 20 / 80 / 400 again but other PE tables ... and
    "odd_lengths": dict(binsizes=[4000, 800, 160], w_max=40000),        # L = 10 / 50 / 250: not multiples of 16 or 64
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
@pytest.mark.parametrize("reg", [False, True])
def test_forward_and_gradients_match_oracle(name, reg):
    from chromoformer_amd import ChromoformerClassifier, ChromoformerRegressor
    cfg = orc._cfg(VARIANTS[name])
    B = 5
    batch = orc.synthetic_batch(B, cfg=cfg, seed=13, regime="realistic", regression=reg)
    P = orc.init_params(cfg, 3, reg)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for v in P.values():
            v.add_(0.05 * torch.randn(v.shape, generator=g))
    Model = ChromoformerRegressor if reg else ChromoformerClassifier
    model = Model(cfg["n_feats"], cfg["d_emb"], cfg["d_head"], cfg["embed"], cfg["pairwise_interaction"], cfg["regulation"],
                  binsizes=cfg["binsizes"], seed=3, i_max=cfg["i_max"], w_max=cfg["w_max"], max_batch=B).cuda(0)
    model.load_state_dict(P)
    for t in P.values():
        t.requires_grad_(True)
    ref_logits = orc.forward(P, batch, cfg)
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
        err = (named[k].grad.cpu() - v.grad).abs().max().item()
        assert err <= GRAD_TOL * v.grad.abs().max().item() + 1e-9, (k, err, v.grad.abs().max().item())
