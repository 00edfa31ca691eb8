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


def test_validation_metrics_equal_the_reference_calls():
    """validation_metrics (numpy) against what train.py:277-318 calls: nn.CrossEntropyLoss / nn.MSELoss (the regressor's
    broadcast [n, n] comparison included), sklearn accuracy / ROC AUC / average precision / r2, scipy pearsonr."""
    import warnings
    from scipy import stats
    from sklearn import metrics
    from chromoformer_amd.train import validation_metrics
    rng = np.random.default_rng(2)
    out = torch.from_numpy(rng.normal(size=(517, 2)).astype(np.float32) * 3)
    lab = torch.from_numpy(rng.integers(0, 2, size=517))
    loss, score, m = validation_metrics(out.numpy(), lab.numpy(), False)
    assert loss.dtype == np.float32 and abs(float(loss) - float(torch.nn.CrossEntropyLoss()(out, lab))) < 1e-6
    ref_score = out.softmax(axis=1)[:, 1].numpy()
    assert score.dtype == ref_score.dtype and np.abs(score - ref_score).max() < 1e-6
    assert abs(m["acc"] - metrics.accuracy_score(lab, out.argmax(axis=1)) * 100) < 1e-9
    assert abs(m["auc"] - metrics.roc_auc_score(lab, ref_score) * 100) < 1e-4 and abs(m["ap"] - metrics.average_precision_score(lab, ref_score) * 100) < 1e-4
    outr = torch.from_numpy(rng.normal(size=(300, 1)).astype(np.float32))
    labr = torch.from_numpy((outr[:, 0] * 0.7 + rng.normal(size=300).astype(np.float32) * 0.5).numpy().astype(np.float32))
    loss, score, m = validation_metrics(outr.numpy(), labr.numpy(), True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref_loss = float(torch.nn.MSELoss()(outr, labr))          # [300, 1] against [300]: the reference's broadcast
    assert abs(float(loss) - ref_loss) < 1e-5 * max(1.0, ref_loss)
    assert score.dtype == np.float32 and np.array_equal(score, outr.flatten().numpy())
    assert abs(m["r2"] - metrics.r2_score(labr.numpy(), outr.flatten().numpy()) * 100) < 1e-3
    assert abs(m["r"] - stats.pearsonr(labr.numpy(), outr.flatten().numpy())[0] * 100) < 1e-4
