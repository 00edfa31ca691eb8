"""Per-stage diagnosis of the HIP path against the restructured CPU statement (oracle/):
prints max-abs error of every named intermediate, forward and backward.  Run on the GPU box:
    python tools/stage_diag.py [B] > gpurun_out/diag.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from chromoformer_amd import ChromoformerClassifier, ChromoformerRegressor
from oracle import chromoformer_oracle as orc
from oracle import restructured as rst


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    reg = len(sys.argv) > 2 and sys.argv[2] == "reg"
    # CF_DIAG_VARIANT=<name of a tests/test_config_variants_gpu.py variant>: another configuration than the default
    cfg = None
    if os.environ.get("CF_DIAG_VARIANT"):
        from tests.test_config_variants_gpu import VARIANTS
        cfg = orc._cfg(VARIANTS[os.environ["CF_DIAG_VARIANT"]])
    batch = orc.synthetic_batch(B, cfg=cfg, seed=3, regime="realistic", regression=reg) if cfg else orc.synthetic_batch(B, seed=3, regime="realistic", regression=reg)
    cls = ChromoformerRegressor if reg else ChromoformerClassifier
    if cfg:
        model = cls(cfg["n_feats"], cfg["d_emb"], cfg["d_head"], cfg["embed"], cfg["pairwise_interaction"], cfg["regulation"],
                    binsizes=cfg["binsizes"], seed=42, i_max=cfg["i_max"], w_max=cfg["w_max"], max_batch=B).cuda(0)
    else:
        model = cls(seed=42, max_batch=B).cuda(0)
    P = orc.init_params(cfg, 42, reg)
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for k, v in P.items():
            v.add_(0.05 * torch.randn(v.shape, generator=g))
    model.load_state_dict(P)
    for t in P.values():
        t.requires_grad_(True)
    keep = {}
    logits_ref = rst.forward(P, batch, cfg, keep=keep)
    for v in keep.values():
        if v.requires_grad:
            v.retain_grad()
    loss_ref = orc.loss_fn(logits_ref, batch["label"], reg)
    loss_ref.backward()

    packed = model.pack_batch(batch)
    logits, loss = model.forward_backward(packed, batch["label"])
    torch.cuda.synchronize()
    print("logits max|d| %.3e   loss %.6f vs %.6f" % ((logits.cpu() - logits_ref.detach()).abs().max().item(), loss.item(), loss_ref.item()))
    S, T = model.i_max, model.i_max + 1
    bins = model.binsizes

    def cmp(name, ref, width=None):
        got = model.debug_buffer(name).cpu()
        ref = ref.detach().reshape(-1)
        if width is not None:  # padded rows
            got = got[: ref.numel() // width[0] * width[1]].view(-1, width[1])[:, : width[0]].reshape(-1)
        got = got[: ref.numel()]
        err = (got - ref).abs().max().item()
        scale = ref.abs().max().item()
        flag = "" if err <= 2e-4 * max(scale, 1e-3) + 1e-6 else "   <<<<<<"
        print("%-18s n=%-8d ref_max %.3e  err %.3e%s" % (name, ref.numel(), scale, err, flag))

    for r, b in enumerate(bins):
        for tag, kt in (("E%d." % r, "E%d." % b),):
            cmp("E%d.x0" % r, keep["E%d.x0" % b])
            for nm in ("q", "qt", "p", "xbar", "a", "y1", "hdn"):
                cmp(tag + nm, keep[kt + nm])
            cmp(tag + "w", keep[kt + "w"], width=(7, 8))
        cmp("P%d.xp0" % r, keep["P%d.xp0" % b])
        for l in range((cfg or orc._cfg(None))["pairwise_interaction"]["n_layers"]):
            tag, kt = "P%d.%d." % (r, l), "P%d.%d." % (b, l)
            for nm in ("q", "qt", "p", "xbar", "a", "y1", "hdn"):
                cmp(tag + nm, keep[kt + nm])
        cmp("R%d.x0" % r, keep["R%d.x0" % b])
        for l in range((cfg or orc._cfg(None))["regulation"]["n_layers"]):
            tag, kt = "R%d.%d." % (r, l), "R%d.%d." % (b, l)
            for nm in ("qkvg", "p", "a", "y1", "hdn"):
                cmp(tag + nm, keep[kt + nm])
            cmp("R%d.x%d" % (r, l + 1), keep[kt + "out"])
    cmp("H.in", keep["H.in"])
    cmp("H.h1", keep["H.h1"])
    print("---- backward intermediates")
    for r, b in enumerate(bins):
        for l in reversed(range((cfg or orc._cfg(None))["regulation"]["n_layers"])):
            tag, kt = "dR%d.%d." % (r, l), "R%d.%d." % (b, l)
            cmp(tag + "a", keep[kt + "a"].grad)
            cmp(tag + "qkvg", keep[kt + "qkvg"].grad)
        cmp("dR%d.x0" % r, keep["R%d.x0" % b].grad)
        for l in reversed(range((cfg or orc._cfg(None))["pairwise_interaction"]["n_layers"])):
            tag, kt = "dP%d.%d." % (r, l), "P%d.%d." % (b, l)
            for nm in ("a", "xbar", "qt", "q"):
                cmp(tag + nm, keep[kt + nm].grad)
        cmp("dP%d.xp0" % r, keep["P%d.xp0" % b].grad)
        tag, kt = "dE%d." % r, "E%d." % b
        for nm in ("a", "xbar", "qt", "q"):
            cmp(tag + nm, keep[kt + nm].grad)
        cmp("dE%d.x" % r, keep["E%d.x0" % b].grad)
    print("---- parameter gradients (vs autograd of the restructured statement)")
    named = dict(model.named_parameters())
    model._publish_grads()
    worst = 0.0
    for k, v in P.items():
        if orc.never_trained(k):
            continue
        got = named[k].grad.cpu()
        scale = v.grad.abs().max().item()
        err = (got - v.grad).abs().max().item()
        rel = err / (scale + 1e-12)
        worst = max(worst, rel)
        if rel > 1e-3:
            print("%-70s scale %.3e err %.3e rel %.2e  <<<<<<" % (k, scale, err, rel))
    print("worst relative-to-max gradient error: %.3e" % worst)


if __name__ == "__main__":
    main()
