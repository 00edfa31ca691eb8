"""Deterministic synthetic Roadmap-style dataset in the reference's on-disk format
(preprocessing/scripts/extract_signals.py:66-71 -> fp16 [7, len] .npy per region;
prepare_train_metadata.py:87-121 -> train.csv columns).  Used where the real data is absent:
golden G6 (reference training run) and the GPU training test regenerate the same files."""
import os

import numpy as np
import pandas as pd


def make_dataset(out_dir, n_genes=48, seed=2024, i_max=8):
    rng = np.random.default_rng(seed)
    os.makedirs(out_dir, exist_ok=True)

    def region(length, level):
        base = np.abs(np.cumsum(rng.normal(0, 0.04, size=(7, length)), axis=1)) * level
        x = rng.poisson(base).astype(np.float16)
        x[:, rng.random(length) < 0.25] = 0
        return x

    rows = []
    for g in range(n_genes):
        chrom = "chr%d" % (1 + g % 4)
        tss = 1_000_000 + 150_000 * g
        label = int(rng.random() < 0.5)
        level = 0.9 if label else 0.25                     # planted signal: expressed genes carry more marks
        np.save(os.path.join(out_dir, "%s:%d-%d.npy" % (chrom, tss - 20000, tss + 20000)), region(40000, level))
        n_part = int(rng.choice([0, 1, 2, 3, 5, 8, 8, 8]))
        names, scores = [], []
        for s in range(min(n_part, i_max)):
            ln = int(np.clip(rng.lognormal(np.log(5900), 0.5), 1800, 40000))
            st = tss + 30_000 + 45_000 * s
            nm = "%s:%d-%d" % (chrom, st, st + ln)
            np.save(os.path.join(out_dir, nm + ".npy"), region(ln, level * 0.8))
            names.append(nm)
            scores.append(round(float(rng.uniform(1.5, 3.0)), 4))
        scores.sort(reverse=True)
        rows.append(dict(gene_id="ENSGSYN%05d" % g, expression=round(float(rng.gamma(2.0, 2.0) * (2.5 if label else 0.2)), 3),
                         eid="E000", label=label, chrom=chrom, start=tss, end=tss + 1, strand="+-"[g % 2], split=1 + g % 4,
                         neighbors=";".join(names) if names else np.nan,
                         scores=";".join(str(s) for s in scores) if scores else np.nan))
    meta = os.path.join(out_dir, "meta.csv")
    pd.DataFrame(rows).to_csv(meta, index=False)
    return meta
