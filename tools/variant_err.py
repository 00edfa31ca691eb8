"""How far the HIP gradients and the fp32 oracle are from an fp64 run of the oracle, per tensor, for variants of tests/test_config_variants_gpu.py:
python tools/variant_err.py [--no64] VARIANT...   (--no64: HIP against the fp32 oracle only)"""
import sys, torch
sys.path.insert(0, "/root/repo")
from oracle import chromoformer_oracle as orc
from chromoformer_amd import ChromoformerClassifier
sys.path.insert(0, "/root/repo/tests")
import importlib
tv = importlib.import_module("tests.test_config_variants_gpu")
NO64 = "--no64" in sys.argv
for name in [a for a in sys.argv[1:] if not a.startswith("--")]:
    cfg = orc._cfg(tv.VARIANTS[name])
    B = 5
    batch = orc.synthetic_batch(B, cfg=cfg, seed=13, regime="realistic", regression=False)
    P = orc.init_params(cfg, 3, False)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for v in P.values():
            v.add_(0.05 * torch.randn(v.shape, generator=g))
    model = ChromoformerClassifier(cfg["n_feats"], cfg["d_emb"], cfg["d_head"], cfg["embed"], cfg["pairwise_interaction"], cfg["regulation"],
                  binsizes=cfg["binsizes"], seed=3, i_max=cfg["i_max"], w_max=cfg["w_max"], max_batch=B).cuda(0)
    model.load_state_dict(P)
    P64 = {k: v.double().clone().requires_grad_(True) for k, v in P.items()}
    for t in P.values():
        t.requires_grad_(True)
    def conv(o):
        if torch.is_tensor(o): return o.double() if o.is_floating_point() else o
        if isinstance(o, (list, tuple)): return type(o)(conv(x) for x in o)
        if isinstance(o, dict): return {k: conv(v) for k, v in o.items()}
        return o
    b64 = conv(batch)
    l32 = orc.loss_fn(orc.forward(P, batch, cfg), batch["label"], False); l32.backward()
    if not NO64:
        l64 = orc.loss_fn(orc.forward(P64, b64, cfg), batch["label"], False); l64.backward()
    logits, loss = model.forward_backward(model.pack_batch(batch), batch["label"])
    model._publish_grads()
    named = dict(model.named_parameters())
    rows = []
    for k, v in P.items():
        if orc.never_trained(k): continue
        ref = v.grad.double() if NO64 else P64[k].grad
        mx = ref.abs().max().item()
        e_hip = (named[k].grad.cpu().double() - ref).abs().max().item() / mx
        e_orc = (v.grad.double() - ref).abs().max().item() / mx
        rows.append((e_hip, e_orc, k))
    rows.sort(reverse=True)
    print(name, "worst 6 (hip vs f64, oracle-f32 vs f64):")
    for r in rows[:6]: print("  %.2e  %.2e  %s" % r)
    rows.sort(key=lambda t: -t[1])
    print(name, "worst 3 of the fp32 oracle against fp64 (a ReLU gate within rounding of zero on the host shows up here as one row of an l1.weight):")
    for r in rows[:3]: print("  %.2e  %.2e  %s" % r)
