"""The committed PMC constants bench.py quotes as `roofline.traffic` (profiles/pmc_traffic.json, profiles/pmc_traffic_stress.json) must parse and
name the kernels the bench line can time: a renamed kernel or a malformed summary would silently turn `traffic` into null."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_step_traffic_constants_cover_the_timed_kernels():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for k in ("k_reg_bwd", "k_reg_fwd", "k_trunk_fwd", "k_trunk_bwd", "k_reduce_opt"):
        assert d["kernels"][k]["hbm_bytes_per_launch"] > 1e7, k
    # five launches of a step: between 0.5 and 2 GB HBM-side together (1.05-1.08 GB measured in rounds 5-6)
    total = sum(d["kernels"][k]["hbm_bytes_per_launch"] for k in ("k_reg_bwd", "k_reg_fwd", "k_trunk_fwd", "k_trunk_bwd", "k_reduce_opt"))
    assert 0.5e9 < total < 2e9


def test_stress_traffic_constants():
    d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_stress.json")))
    assert set(d["kernels"]) >= {"k_attn_fwd", "k_attn_delta", "k_attn_bwd2"}
    # compulsory bytes of the stress shape: 8 tensors of 0.891 GB; the 128-key backward moves less than 30 GB, the 64-key one moved 48
    assert 7e9 < d["kernels"]["k_attn_fwd"] + d["kernels"]["k_attn_delta"] + d["kernels"]["k_attn_bwd2"] < 45e9
    assert d["kernels"]["k_attn_bwd2"] < 30e9
