"""The numpy running-metric helpers of the training loop equal the sklearn / scipy functions the reference calls
(train.py:205-232), ties included."""
import numpy as np
import pytest
import torch


def test_binary_auc_ap_equal_sklearn():
    from sklearn import metrics
    from chromoformer_amd.train import binary_auc_ap
    rng = np.random.default_rng(0)
    for n, ties in ((640, False), (640, True), (17, True), (3, False)):
        y = rng.integers(0, 2, size=n)
        y[0], y[1] = 0, 1
        s = rng.random(n)
        if ties:
            s = np.round(s, 1)
        auc, ap = binary_auc_ap(y, s)
        assert abs(auc - metrics.roc_auc_score(y, s)) < 1e-12
        assert abs(ap - metrics.average_precision_score(y, s)) < 1e-12
    with pytest.raises(ValueError):
        binary_auc_ap(np.ones(8), rng.random(8))


def test_report_lines_equal_the_reference_formulas():
    from scipy import stats
    from sklearn import metrics
    from chromoformer_amd.train import _report_train
    rng = np.random.default_rng(1)
    lines, logs = [], []
    wb = type("W", (), {"log": staticmethod(logs.append)})
    out, lab = torch.from_numpy(rng.normal(size=(80, 2)).astype(np.float32)), torch.from_numpy(rng.integers(0, 2, size=80))
    _report_train(lines.append, wb, 3, 0.5, 3e-5, out, lab, False)
    score, pred = out.softmax(axis=1)[:, 1], out.argmax(axis=1)
    want = (metrics.accuracy_score(lab, pred) * 100, metrics.roc_auc_score(lab, score) * 100, metrics.average_precision_score(lab, score) * 100)
    got = logs[-1]
    assert abs(got["train/acc"] - want[0]) < 1e-9 and abs(got["train/auc"] - want[1]) < 1e-9 and abs(got["train/ap"] - want[2]) < 1e-9
    assert lines[-1].startswith("E3 0.5000, lr=3e-05, acc=")
    outr, labr = torch.from_numpy(rng.normal(size=(80, 1)).astype(np.float32)), torch.from_numpy(rng.normal(size=80).astype(np.float32))
    _report_train(lines.append, wb, 3, 0.5, 3e-5, outr, labr, True)
    got = logs[-1]
    assert abs(got["train/r2"] - metrics.r2_score(labr.flatten(), outr.flatten()) * 100) < 1e-3      # sklearn / scipy work in the float32 of their inputs
    assert abs(got["train/r"] - stats.pearsonr(labr.flatten(), outr.flatten())[0] * 100) < 1e-6
