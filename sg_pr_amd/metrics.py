"""PR / ROC / F1-max of the evaluation loop (eval_batch.py:48-49, 69, 85-87) in numpy.

Same definitions as the sklearn functions the reference calls
(`precision_recall_curve`, `roc_curve`, `auc`); float64 throughout.
"""
import numpy as np


def _binary_clf_curve(gt, score):
    gt = np.asarray(gt, dtype=np.float64).ravel()
    score = np.asarray(score).ravel()
    if gt.size != score.size:
        raise ValueError("gt and score differ in length")
    order = np.argsort(score, kind="mergesort")[::-1]
    s, y = score[order], gt[order]
    ends = np.r_[np.where(np.diff(s))[0], y.size - 1]
    tps = np.cumsum(y)[ends]
    fps = 1 + ends - tps
    return fps, tps, s[ends]


def precision_recall_curve(gt, score):
    fps, tps, thr = _binary_clf_curve(gt, score)
    ps = tps + fps
    precision = np.zeros_like(tps)
    np.divide(tps, ps, out=precision, where=(ps != 0))
    recall = np.ones_like(tps) if tps[-1] == 0 else tps / tps[-1]
    return np.hstack((precision[::-1], 1.0)), np.hstack((recall[::-1], 0.0)), thr[::-1]


def f1_max(gt, score):
    """eval_batch.py:85-87: F1 = 2PR/(P+R), nan_to_num, max."""
    p, r, _ = precision_recall_curve(gt, score)
    with np.errstate(divide="ignore", invalid="ignore"):
        f1 = 2 * p * r / (p + r)
    return float(np.max(np.nan_to_num(f1)))


def roc_auc(gt, score):
    """eval_batch.py:48-49 (roc_curve without drop_intermediate does not change the area)."""
    fps, tps, _ = _binary_clf_curve(gt, score)
    fps = np.r_[0.0, fps]
    tps = np.r_[0.0, tps]
    if fps[-1] <= 0 or tps[-1] <= 0:
        return float("nan")
    trapezoid = getattr(np, "trapezoid", None) or np.trapz
    return float(trapezoid(tps / tps[-1], fps / fps[-1]))
