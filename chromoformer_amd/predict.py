"""`python -m chromoformer_amd.predict` -- the inference entrypoint (reference: demo/run_demo.py and
demo/run_demo_regression.py, same -m / -d / -o / -w options plus --regression instead of a second script).

    prediction column = sigmoid(logits)[:, 1]   (classifier, demo/run_demo.py:116)
                      = logits[:, 0]            (regressor,  demo/run_demo_regression.py:117)

Forward only (`cf_forward(..., save_for_backward=0)`) on batches staged from a GeneStore.  Checkpoints in the
reference's `.pt` layout load directly; checkpoints written before the reference renamed its modules
(`embed2000_a`, `transformer500`, `lin_proj_c`, ... -- the mapping of misc/convert_weight.py:19-88) are renamed
on the fly."""
from __future__ import annotations

import argparse
import re

import numpy as np
import pandas as pd
import torch

from .data import ChromoformerDataset, GeneStore
from .engine import Slot, Trainer
from .net import ChromoformerClassifier, ChromoformerRegressor
from .util import seed_everything

_LEGACY = [  # (pattern, replacement), applied in order; first match of the second group wins (convert_weight.py)
    (r"transformer(2000|500|100)", r"regulation.\1.transformer"),
    (r"embed(2000|500|100)_a", r"embed.\1"),
    (r"embed(2000|500|100)_b", r"pairwise_interaction.\1"),
    (r"^embed(?=\d)", "embed."),
    (r"^pw_int", "pairwise_interaction."),
    (r"^reg(?=\d)", "regulation."),
]


def modernise_keys(state):
    """Legacy checkpoint keys -> current `state_dict` keys (no-op for current checkpoints)."""
    out = {}
    for k, v in state.items():
        k = k.replace("lin_proj_c.", "lin_proj_pcre.") if "lin_proj_c." in k else k
        for pat, rep in _LEGACY:
            k2 = re.sub(pat, rep, k, count=1)
            if k2 != k:
                k = k2
                break
        out[k] = v
    return out


def predict(meta_path, npy_dir, weights=None, regression=False, bsz=32, seed=123, i_max=8, w_prom=40000, w_max=40000,
            binsizes=(2000, 500, 100), progress=False, store_path=None):
    """-> (meta DataFrame, predictions float32 [n_genes]) in the order of the metadata file."""
    seed_everything(seed)
    meta = pd.read_csv(meta_path)
    genes = meta.gene_id.tolist()
    from . import pack
    packed = pack.find(npy_dir, store_path, list(binsizes), i_max, w_prom, w_max, 7, genes, meta=meta)
    dev = torch.device("cuda", torch.cuda.current_device())
    if packed is not None:
        store = packed.store(genes, device=dev, regression=False)          # labels are not used for prediction
    else:
        ds = ChromoformerDataset(meta_path, npy_dir, genes, 7, i_max, list(binsizes), w_prom, w_max, regression=regression)
        store = GeneStore(ds, progress=progress, device=dev, resident=True)      # binned on the GPU, resident (126 KB per gene)
    Model = ChromoformerRegressor if regression else ChromoformerClassifier
    model = Model(7, 128, 128, dict(n_layers=1, n_heads=2, d_model=128, d_ff=128), dict(n_layers=2, n_heads=2, d_model=128, d_ff=256),
                  dict(n_layers=6, n_heads=8, d_model=256, d_ff=256), binsizes=list(binsizes), seed=seed, i_max=i_max, w_max=w_max,
                  max_batch=bsz)
    if weights is not None:
        ckpt = torch.load(weights, map_location="cpu", weights_only=False)
        model.load_state_dict(modernise_keys(ckpt["net"] if "net" in ckpt else ckpt))
    model.cuda()
    trainer = Trainer(model, use_graph=False)
    out = trainer.evaluate_store(store, bsz).cpu()          # device gather + forward per batch, logits in store order
    preds = [out.numpy().reshape(-1) if regression else torch.sigmoid(out).numpy()[:, 1]]
    return meta, np.concatenate(preds).astype(np.float32)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Chromoformer expression prediction on the MI355X path")
    ap.add_argument("-m", "--meta", required=True, help="Path to input metadata file.")
    ap.add_argument("-d", "--npy-dir", required=True, help="Path to directory containing histone signals in .npy files.")
    ap.add_argument("-o", "--output", required=True, help="Path to output expression prediction.")
    ap.add_argument("-w", "--weights", default=None, help="Path to pretrained Chromoformer weights in .pt format.")
    ap.add_argument("--regression", action="store_true", help="ChromoformerRegressor (run_demo_regression.py)")
    ap.add_argument("--store", default=None, help="packed store of `python -m chromoformer_amd.pack` (default: <npy-dir>/chromoformer.cfstore if present)")
    args = ap.parse_args(argv)
    meta, pred = predict(args.meta, args.npy_dir, args.weights, args.regression, progress=True, store_path=args.store)
    print("Predicting expressions for %d genes." % len(meta))
    meta["prediction"] = pred
    meta.to_csv(args.output, index=False)
    from sklearn import metrics
    if args.regression:
        from scipy.stats import pearsonr
        y = np.log2(meta["expression"] + 1)
        print("R2 : %s" % metrics.r2_score(y, meta["prediction"]))
        print("Pearson's r : %s" % pearsonr(y, meta["prediction"])[0])
    else:
        print("ROC-AUC : %s" % metrics.roc_auc_score(meta["label"], meta["prediction"]))
        print("Average Precision : %s" % metrics.average_precision_score(meta["label"], meta["prediction"]))
        print("Accuracy : %s" % metrics.accuracy_score(meta["label"], (meta["prediction"] > 0.5).astype(int)))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
