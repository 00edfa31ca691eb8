"""The benchmark's own size (bsz 64, default configuration, dense synthetic inputs) against the oracle, and the independent
kernel implementations of the library against each other at that size.

Small-batch tests run the centre-row attention with one region per workgroup (k_attc1) in both stages and leave most CUs
idle; at bsz 64 the Pairwise launches take the eight-regions-per-workgroup MFMA kernel (k_attc2<., 8>), the Regulation kernels
run 192 workgroups and the weight-gradient tables are fully populated.  The oracle's forward + autograd of one 64-gene batch
takes a few seconds on the host.  Switches (environment, read when a model is constructed): CF_ATTC1=0 puts the one-region
launches back on the MFMA kernel, CF_REG_FUSED=0 runs the Regulation stack layer by layer on the stand-alone kernels instead of the fused 512-thread ones, CF_HEAD_RIDE=0 the prediction head as a launch of its own (matrix cores, 16 genes per
workgroup) instead of at the tail of the Regulation forward launch (vector ALUs, one gene per workgroup), CF_REG_ROW0=0 the last Regulation
layer over all T rows instead of the one row the head consumes (cf_reg8.h, b_run_row0) -- different code, same math."""
import os

import pytest
import torch

from oracle import chromoformer_oracle as orc

pytestmark = pytest.mark.gpu
B = 64
# Gradients at this size: per tensor, relative Frobenius error <= 1e-3 and largest entry-wise error <= 1e-2 of the largest entry.
# Two effects set the scale, both in the oracle as much as here (measured: worst tensor 3.7e-4 / 2.2e-3): the softmax backward
# p (dp - <p, dp>) cancels almost completely when 400 bins attend nearly uniformly, which lifts fp32 rounding to ~2e-4 relative
# for every tensor behind the 100-bp Pairwise stack; and among the 2.6 million ReLU gates of a 64-gene batch a few sit within
# rounding of zero -- one gate open on one side and shut on the other changes a row of a weight gradient by ~1e-5 absolute.
LOGIT_TOL, GRAD_FROB, GRAD_MAX = 1e-4, 1e-3, 1e-2


def _close(g, ref):
    return ((g - ref).norm() <= GRAD_FROB * ref.norm() + 1e-12) and ((g - ref).abs().max() <= GRAD_MAX * ref.abs().max() + 1e-9)


def _model_run(env, batch, P):
    from chromoformer_amd import ChromoformerClassifier
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        model = ChromoformerClassifier(seed=42, max_batch=B).cuda(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    model.load_state_dict(P)
    logits, loss = model.forward_backward(model.pack_batch(batch), batch["label"])
    torch.cuda.synchronize()
    model._publish_grads()
    grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
    return logits.cpu().clone(), float(loss), grads


@pytest.fixture(scope="module")
def case():
    batch = orc.synthetic_batch(B, seed=2024, regime="dense")
    P = orc.init_params(None, 42, False)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for v in P.values():
            v.add_(0.02 * torch.randn(v.shape, generator=g))
    return batch, P


def test_bsz64_forward_loss_and_all_gradients_match_the_oracle(case):
    batch, P = case
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    ref_logits = orc.forward(Pr, batch)
    ref_loss = orc.loss_fn(ref_logits, batch["label"], False)
    ref_loss.backward()
    logits, loss, grads = _model_run({}, batch, P)
    assert (logits - ref_logits.detach()).abs().max() < LOGIT_TOL
    assert abs(loss - ref_loss.item()) < 1e-4 * max(1.0, abs(ref_loss.item()))
    n = 0
    for k, v in Pr.items():
        if orc.never_trained(k):
            assert k not in grads
            continue
        assert _close(grads[k], v.grad), (k, ((grads[k] - v.grad).norm() / v.grad.norm()).item(), (grads[k] - v.grad).abs().max().item())
        n += 1
    assert n == 334


@pytest.mark.parametrize("env", [{"CF_ATTC1": "0"}, {"CF_REG_FUSED": "0"}, {"CF_HEAD_RIDE": "0"}, {"CF_REG_ROW0": "0"}],
                         ids=["attc1_vs_attc2", "fused_regulation_vs_layer_by_layer", "head_ride_vs_head_launch", "last_layer_row0_vs_all_rows"])
def test_bsz64_independent_kernel_implementations_agree(case, env):
    batch, P = case
    logits, loss, grads = _model_run({}, batch, P)
    logits2, loss2, grads2 = _model_run(env, batch, P)
    assert (logits - logits2).abs().max() < 2e-5
    assert abs(loss - loss2) < 2e-6 * max(1.0, abs(loss))
    assert grads.keys() == grads2.keys()
    for k in grads:
        assert _close(grads2[k], grads[k]), (k, ((grads[k] - grads2[k]).norm() / grads[k].norm()).item())


def test_bsz64_fused_trunk_equals_the_stand_alone_kernels(case):
    """The fused centre-row trunk (cf_trunk.h: Embedding + Pairwise stage of a gene in one workgroup, one launch per direction)
    against the stand-alone kernels (CF_TRUNK=0).  The Pairwise phases ARE the stand-alone kernels' bodies; the Embedding layer's
    one-row chains run on the vector ALUs in the trunk (cf_trunk_e.h: the same products, another summation order), so the two
    paths agree to fp32 rounding, not bit for bit: logits 2e-6, every gradient 1e-5 of its largest element (the Regulation and
    head gradients see the Embedding row only through the forward pass)."""
    batch, P = case
    logits, loss, grads = _model_run({}, batch, P)
    logits2, loss2, grads2 = _model_run({"CF_TRUNK": "0"}, batch, P)
    assert (logits - logits2).abs().max() <= 2e-6 and abs(loss - loss2) <= 1e-6, ((logits - logits2).abs().max().item(), loss, loss2)
    assert grads.keys() == grads2.keys()
    for k in grads:
        assert _close(grads2[k], grads[k]), (k, ((grads[k] - grads2[k]).norm() / grads[k].norm()).item())
        assert (grads[k] - grads2[k]).abs().max() <= 1e-5 * grads[k].abs().max() + 1e-12, (k, ((grads[k] - grads2[k]).abs().max() / grads[k].abs().max()).item())
