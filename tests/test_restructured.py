"""The restructured (centre-row, weight-absorbed) algorithm the HIP kernels execute is
the same function as the dense oracle: forward and all 334 live gradients, on CPU."""
import numpy as np
import torch

from oracle import chromoformer_oracle as orc
from oracle import restructured as rst
from tests.helpers import load_npz_batch, take


def _grads(fn, P, batch, label, reg):
    for t in P.values():
        t.grad = None
        t.requires_grad_(True)
    logits = fn(P, batch)
    orc.loss_fn(logits, label, reg).backward()
    return logits.detach(), {k: (None if v.grad is None else v.grad.clone()) for k, v in P.items()}


def test_restructured_equals_dense_demo_and_kat():
    for name, seed in (("demo_subset.npz", 123), ("kat.npz", 42)):
        batch, ex = load_npz_batch(name)
        if name == "kat.npz":
            batch = take(batch, [0, 1, 2])
        P = orc.init_params(None, seed, False)
        with torch.no_grad():
            d = (rst.forward(P, batch) - orc.forward(P, batch)).abs().max().item()
        assert d < 2e-6, (name, d)


def test_restructured_gradients_match_dense():
    batch = orc.synthetic_batch(3, seed=5, regime="realistic")
    for reg in (False, True):
        P = orc.init_params(None, 42, reg)
        with torch.no_grad():   # move away from init so that LN gains / biases are generic
            g = torch.Generator().manual_seed(1)
            for k, v in P.items():
                v.add_(0.05 * torch.randn(v.shape, generator=g))
        label = torch.tensor([1.5, 0.0, 3.0]) if reg else torch.tensor([1, 0, 1])
        l1, g1 = _grads(orc.forward, P, batch, label, reg)
        l2, g2 = _grads(rst.forward, P, batch, label, reg)
        assert (l1 - l2).abs().max() < 5e-6
        for k in P:
            if orc.never_trained(k):
                assert g2[k] is None or float(g2[k].abs().max()) == 0.0
                continue
            scale = float(g1[k].abs().max()) + 1e-12
            err = float((g1[k] - g2[k]).abs().max()) / scale
            assert err < 2e-4, (k, err)
