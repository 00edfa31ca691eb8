"""The product-side benchmark workload generator (chromoformer_amd/synth.py) produces exactly the batches of the oracle's
generator for the same seed: both regimes, classifier / regressor labels, a non-default partner count."""
import torch

from chromoformer_amd.synth import synthetic_batch
from oracle import chromoformer_oracle as orc


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        if isinstance(a[k], dict):
            assert list(a[k]) == list(b[k])
            for r in a[k]:
                assert a[k][r].dtype == b[k][r].dtype and torch.equal(a[k][r], b[k][r]), (k, r)
        else:
            assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k


def test_generator_matches_the_oracle_generator():
    for regime in ("dense", "realistic"):
        for reg in (False, True):
            _same(synthetic_batch(5, seed=77, regime=regime, regression=reg), orc.synthetic_batch(5, seed=77, regime=regime, regression=reg))
    _same(synthetic_batch(3, seed=5, regime="realistic", i_max=4), orc.synthetic_batch(3, cfg=orc._cfg(dict(i_max=4)), seed=5, regime="realistic"))
