"""Shared helpers for the tests: golden loading and batch plumbing."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BATCH_KEYS = ("promoter_feats", "promoter_pad_masks", "pcre_feats", "pcre_pad_masks", "interaction_masks")


def load_npz_batch(name):
    """-> (batch dict in the reference layout, dict of the remaining arrays)."""
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    batch = {k: {} for k in BATCH_KEYS}
    extra = {}
    for key in z.files:
        head, _, tail = key.partition(".")
        if head in BATCH_KEYS:
            batch[head][int(tail)] = torch.from_numpy(z[key])
        elif key in ("interaction_freq", "label"):
            batch[key] = torch.from_numpy(z[key])
        else:
            extra[key] = z[key]
    return batch, extra


def take(batch, idx):
    return {k: ({b: t[idx] for b, t in v.items()} if isinstance(v, dict) else v[idx]) for k, v in batch.items()}


def checksum(t):
    t = t.detach().double()
    return np.array([float(t.sum()), float(t.abs().sum()), float((t * t).sum())])
