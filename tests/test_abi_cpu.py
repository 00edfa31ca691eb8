"""Host-only checks of the C-ABI library: it loads, exports every symbol of
include/chromoformer_hip.h, and its parameter table equals the reference's state_dict
layout (golden G3).  No compute call is made (there is no GPU here)."""
import json
import os
import re

import torch

from chromoformer_amd import _lib
from tests.helpers import GOLDEN

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "chromoformer_hip.h")).read()
    declared = set(re.findall(r"\b(cf_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.cf_abi_version() == 1


def _cfg(n_out=2):
    return _lib.make_config(7, 128, 128, n_out, [2000, 500, 100], [20, 80, 400], 8,
                            {"n_layers": 1, "n_heads": 2, "d_model": 128, "d_ff": 128},
                            {"n_layers": 2, "n_heads": 2, "d_model": 128, "d_ff": 256},
                            {"n_layers": 6, "n_heads": 8, "d_model": 256, "d_ff": 256}, 64)


def test_param_table_matches_reference_state_dict():
    g = json.load(open(os.path.join(GOLDEN, "state_dict.json")))
    for n_out, key in ((2, "seed42_clf"), (1, "seed42_reg")):
        lay, tab = _lib.param_layout(_cfg(n_out))
        assert [t["name"] for t in tab] == g[key]["keys"]
        assert [list(t["shape"]) for t in tab] == g[key]["shapes"]
        assert lay.n_tensors == 370
        assert sum(not t["trainable"] for t in tab) == 36
        assert sum(t["numel"] for t in tab if not t["trainable"]) == 1086
        # trainable tensors first, contiguous, 16-byte aligned, non-overlapping
        spans = sorted((t["offset"], t["offset"] + t["numel"], t["trainable"]) for t in tab)
        for (a0, a1, _), (b0, _, _) in zip(spans, spans[1:]):
            assert a1 <= b0
        assert all(t["offset"] % 4 == 0 for t in tab)
        assert max(t["offset"] + t["numel"] for t in tab if t["trainable"]) <= lay.n_active
        assert min(t["offset"] for t in tab if not t["trainable"]) >= lay.n_active
    assert lay.n_elems + 129 == g["n_params"] == 5342672


def test_unsupported_configs_are_rejected_loudly():
    import ctypes as C
    cfg = _cfg()
    cfg.embed_layers = 9            # up to 8 Embedding layers are supported (more than one runs the all-rows path)
    lay = _lib.cf_layout()
    assert _lib.lib().cf_param_layout(C.byref(cfg), C.byref(lay), None, 0) != 0
    assert b"embed.n_layers" in _lib.lib().cf_last_error()
    # widths the reference leaves free (net.py:277-278) and this library does not implement: refused by name, with the reference site
    # (d_head: any multiple of 4 up to 1024 -- 128 on the matrix cores, other widths on the vector ALUs; d_emb: 64, 128, 256 since round 5)
    for field, value, site in (("d_emb", 192, b"net.py:277"), ("d_emb", 512, b"net.py:277"), ("d_head", 130, b"net.py:278"), ("d_head", 2048, b"net.py:278")):
        cfg = _cfg()
        setattr(cfg, field, value)
        assert _lib.lib().cf_param_layout(C.byref(cfg), C.byref(lay), None, 0) != 0
        err = _lib.lib().cf_last_error()
        assert field.encode() in err and site in err and b"not supported" in err
    cfg = _cfg()
    cfg.d_head = 96
    assert _lib.lib().cf_param_layout(C.byref(cfg), C.byref(lay), None, 0) == 0
    # d_emb = 64 / 256: the Embedding and the Pairwise widths follow it (net.py:305, 361-370), one Embedding layer, at most two heads there
    for d in (64, 256):
        cfg = _cfg()
        cfg.d_emb = cfg.embed_dmodel = cfg.pair_dmodel = d
        assert _lib.lib().cf_param_layout(C.byref(cfg), C.byref(lay), None, 0) == 0, _lib.lib().cf_last_error()
        cfg.pair_dmodel = 128                                                      # the two widths must agree
        assert _lib.lib().cf_param_layout(C.byref(cfg), C.byref(lay), None, 0) != 0 and b"d_model = d_emb" in _lib.lib().cf_last_error()
        cfg.pair_dmodel, cfg.embed_heads = d, 4
        assert _lib.lib().cf_param_layout(C.byref(cfg), C.byref(lay), None, 0) != 0 and b"at most two heads" in _lib.lib().cf_last_error()
    # head counts: Embedding / Pairwise 1, 2, 4 (d_model = 128), Regulation 1, 2, 4, 8 or 16 heads (round 6) with d_model 128 or 256; the rest is refused by name
    for field, value, ok in (("embed_heads", 1, True), ("embed_heads", 4, True), ("embed_heads", 8, False), ("pair_heads", 4, True),
                             ("pair_heads", 3, False), ("pair_dmodel", 256, False), ("reg_heads", 4, True), ("reg_heads", 2, True), ("reg_heads", 16, True), ("reg_heads", 3, False), ("reg_heads", 32, False),
                             ("reg_dmodel", 128, True), ("reg_dmodel", 512, False)):
        cfg = _cfg()
        setattr(cfg, field, value)
        rc = _lib.lib().cf_param_layout(C.byref(cfg), C.byref(lay), None, 0)
        assert (rc == 0) == ok, (field, value)
        if not ok:
            assert b"n_heads in" in _lib.lib().cf_last_error()
    cfg = _cfg()
    cfg.embed_heads, cfg.embed_layers = 4, 2       # the all-rows Embedding path (n_layers > 1) exists for two heads only
    assert _lib.lib().cf_param_layout(C.byref(cfg), C.byref(lay), None, 0) != 0 and b"all-rows" in _lib.lib().cf_last_error()


def test_model_state_dict_and_seeded_init_match_golden():
    from chromoformer_amd import ChromoformerClassifier, ChromoformerRegressor
    from tests.helpers import checksum
    import numpy as np
    g = json.load(open(os.path.join(GOLDEN, "state_dict.json")))
    for cls, tag in ((ChromoformerClassifier, "clf"), (ChromoformerRegressor, "reg")):
        for seed in (42, 123):
            ref = g["seed%d_%s" % (seed, tag)]
            sd = cls(seed=seed).state_dict()
            assert list(sd.keys()) == ref["keys"]
            got = np.array([checksum(v) for v in sd.values()])
            np.testing.assert_allclose(got, np.array(ref["checksums"]), rtol=1e-12, atol=1e-12)


def test_no_gpu_means_loud_failure_not_fallback():
    from chromoformer_amd import ChromoformerClassifier
    if torch.cuda.is_available():
        return
    m = ChromoformerClassifier()
    try:
        m.cuda()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("expected a RuntimeError without a GPU")
