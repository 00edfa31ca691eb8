"""Inference entrypoint: legacy checkpoint keys (host only) and, on the GPU, `python -m chromoformer_amd.predict`
against the oracle's forward on the same synthetic dataset."""
import os

import numpy as np
import pytest
import torch

from oracle import chromoformer_oracle as orc
from tests.synth_data import make_dataset


def test_legacy_checkpoint_keys_map_onto_the_current_state_dict():
    from chromoformer_amd.predict import modernise_keys
    cur = [e[0] for e in orc.param_spec()]
    legacy = []
    for k in cur:
        k = k.replace("lin_proj_pcre.", "lin_proj_c.")
        for b in ("2000", "500", "100"):
            k = k.replace("regulation.%s.transformer" % b, "transformer" + b)
            k = k.replace("pairwise_interaction.%s" % b, "embed%s_b" % b).replace("embed.%s" % b, "embed%s_a" % b)
        legacy.append(k)
    assert legacy != cur
    assert list(modernise_keys({k: 0 for k in legacy})) == cur
    assert list(modernise_keys({k: 0 for k in cur})) == cur          # current checkpoints pass through untouched
    older = {"embed2000.lin_proj.weight": 0, "pw_int500.lin_proj_p.weight": 0, "reg100.transformer.layers.0.ff.l1.bias": 0}
    assert list(modernise_keys(older)) == ["embed.2000.lin_proj.weight", "pairwise_interaction.500.lin_proj_p.weight",
                                           "regulation.100.transformer.layers.0.ff.l1.bias"]


@pytest.mark.gpu
@pytest.mark.parametrize("reg", [False, True])
def test_predict_matches_oracle_forward(tmp_path, reg):
    from chromoformer_amd import predict
    from chromoformer_amd.data import ChromoformerDataset
    meta = make_dataset(str(tmp_path / "npy"), n_genes=20, seed=11)
    P = orc.init_params(seed=7, regression=reg)
    ck = str(tmp_path / "w.pt")
    torch.save({"net": P}, ck)
    out = str(tmp_path / "pred.csv")
    argv = ["-m", meta, "-d", str(tmp_path / "npy"), "-o", out, "-w", ck] + (["--regression"] if reg else [])
    assert predict.main(argv) == 0
    import pandas as pd
    got = pd.read_csv(out)["prediction"].to_numpy()
    ds = ChromoformerDataset(meta, str(tmp_path / "npy"), pd.read_csv(meta).gene_id.tolist(), regression=reg)   # pinned to the reference by G5
    batch = torch.utils.data.default_collate([ds[i] for i in range(len(ds))])
    with torch.no_grad():
        logits = orc.forward(P, batch)
    ref = logits.numpy().reshape(-1) if reg else torch.sigmoid(logits).numpy()[:, 1]
    assert np.abs(got - ref).max() < 2e-5
