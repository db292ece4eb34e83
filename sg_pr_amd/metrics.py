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


def roc_curve(gt, score, drop_intermediate=True):
    """eval_batch.py:48: sklearn.metrics.roc_curve's definition -> (fpr, tpr, thresholds).  Collinear points are
    dropped like sklearn's default does; the first point is (0, 0) at threshold +inf."""
    fps, tps, thr = _binary_clf_curve(gt, score)
    if drop_intermediate and fps.size > 2:
        keep = np.where(np.r_[True, np.logical_or(np.diff(fps, 2), np.diff(tps, 2)), True])[0]
        fps, tps, thr = fps[keep], tps[keep], thr[keep]
    fps = np.r_[0.0, fps]
    tps = np.r_[0.0, tps]
    thr = np.r_[np.inf, thr]
    fpr = fps / fps[-1] if fps[-1] > 0 else np.full(fps.shape, np.nan)
    tpr = tps / tps[-1] if tps[-1] > 0 else np.full(tps.shape, np.nan)
    return fpr, tpr, thr


def auc(x, y):
    """sklearn.metrics.auc for a monotone x: trapezoidal area (eval_batch.py:49)."""
    trapezoid = getattr(np, "trapezoid", None) or np.trapz
    return float(trapezoid(y, x))


def roc_auc(gt, score):
    """eval_batch.py:48-49: area under the ROC curve (dropping collinear points does not change it)."""
    fpr, tpr, _ = roc_curve(gt, score, drop_intermediate=False)
    if np.isnan(fpr).any() or np.isnan(tpr).any():
        return float("nan")
    return auc(fpr, tpr)


# ---------------------------------------------------------------------------------------------------------------------
# Device-side F1-max and ROC area: exact, without sorting the matrix.
#
# F1(t) = 2 TP / (TP + P + FP) can only peak at a threshold t that is the score of a POSITIVE pair (moving t down to a
# negative-only value adds false positives and nothing else), and positives are rare (loop closures).  So the engine
# hands over the scores of the positive pairs (sgpr_pair_positives); their sorted distinct values u[0..U) give TP(>= u[i])
# at once, and one streaming pass over the matrix (sgpr_pair_threshold_counts) counts the negatives between up to 8191
# thresholds: every S-th distinct value gets its exact FP, the S - 1 values between two thresholds a bound (FP is at
# least that of the next threshold).  Segments whose bound beats the best exact F1 are settled by a second pass that
# takes their values as thresholds.  The same first pass ranks every negative among all u (Mann-Whitney):
# AUC = (#{pos > neg} + #{pos == neg} / 2) / (P N), which is sklearn's trapezoid area, ties included.
# `count_fn(thresholds, rank)` -> (int64 [T + 1] negatives with exactly b thresholds <= score, rank_sum or None);
# rank = (u, S, above) asks for rank_sum = sum over negatives of 2 #{positive pairs > s} + #{positive pairs == s}.
MAX_THRESHOLDS = 8191


def _f1(tp, fp, pos):
    tp = np.asarray(tp, dtype=np.float64)
    fp = np.asarray(fp, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = np.where(tp + fp > 0, tp / (tp + fp), 0.0)
        r = tp / pos if pos > 0 else np.ones_like(tp)
        f = 2 * p * r / (p + r)
    return np.nan_to_num(f)


def distinct_counts(pos_scores):
    """Ascending distinct values of the positive scores and how many pairs carry each."""
    u, mult = np.unique(np.asarray(pos_scores, dtype=np.float32).ravel(), return_counts=True)
    return u, mult.astype(np.int64)


def pr_roc_from_counts(pos_scores, count_fn, want_auc=True, max_thresholds=MAX_THRESHOLDS, distinct=None, refine=True):
    """(F1-max of eval_batch.py:85-87, ROC area of eval_batch.py:48-49, counting passes) from the scores of the
    positive pairs (or `distinct` = their ascending distinct values and multiplicities) and a counting function over
    the negatives (see above).  Without positives F1-max is 0 and the area NaN, like the sorted path (metrics.f1_max /
    roc_auc).  refine=False stops after the first pass (the area is exact then, F1-max a lower bound)."""
    u, mult = distinct if distinct is not None else distinct_counts(pos_scores)
    n_u, p_all = int(u.size), int(mult.sum())
    if n_u == 0:
        return 0.0, float("nan"), 0
    above = np.concatenate((np.cumsum(mult[::-1])[::-1], [0])).astype(np.int64)   # positive pairs with value >= u[i]
    # interval = values [lo, hi) whose FP is not known yet, with a lower bound of it (the exact FP of value hi, 0 past
    # the end); F1 inside it is at most F1(TP(>= u[lo]), that bound).  A pass takes whole intervals as thresholds, best
    # bound first, as many as fit - or, when the best one alone exceeds the budget, as the initial [0, U) may, every
    # stride-th value of it, which leaves the values in between as new intervals.
    ilo, ihi, ifp = np.array([0]), np.array([n_u]), np.array([0])
    best, auc, neg, passes = -1.0, float("nan"), None, 0
    while True:
        bound = _f1(above[ilo], ifp, p_all)
        keep = bound > best
        if not keep.any():
            break
        ilo, ihi, ifp, bound = ilo[keep], ihi[keep], ifp[keep], bound[keep]
        order = np.argsort(-bound, kind="stable")
        ilo, ihi, ifp = ilo[order], ihi[order], ifp[order]
        lens = ihi - ilo
        ntake = int(np.searchsorted(np.cumsum(lens), max_thresholds, side="right"))
        if ntake == 0:                                                   # the best interval alone is too long: sample it
            fine = np.arange(ilo[0], ihi[0], -(-int(lens[0]) // max_thresholds))
        else:
            ln = lens[:ntake]
            fine = np.sort(np.repeat(ilo[:ntake] - (np.cumsum(ln) - ln), ln) + np.arange(int(ln.sum())))
        rank = None
        if passes == 0 and want_auc:                                     # the first pass samples [0, U) from 0
            rank = (u, int(fine[1] - fine[0]) if fine.size > 1 else n_u, above)
        counts, rank_sum = count_fn(u[fine], rank)
        counts = np.asarray(counts, dtype=np.int64)
        passes += 1
        if neg is None:
            neg = int(counts.sum())
            if rank is not None and neg > 0:
                auc = float(rank_sum) / (2.0 * p_all * neg)
        fp = neg - np.cumsum(counts)[:-1]                                # FP(score >= u[fine[q]]), exact
        best = max(best, float(_f1(above[fine], fp, p_all).max()))
        if ntake == 0:                                                   # the gaps of the sampled interval stay open
            g_lo = np.concatenate(([ilo[0]], fine + 1))
            g_hi = np.concatenate((fine, [ihi[0]]))
            g_fp = np.concatenate((fp, [ifp[0]]))
            open_ = g_hi > g_lo
            ilo = np.concatenate((ilo[1:], g_lo[open_]))
            ihi = np.concatenate((ihi[1:], g_hi[open_]))
            ifp = np.concatenate((ifp[1:], g_fp[open_]))
        else:
            ilo, ihi, ifp = ilo[ntake:], ihi[ntake:], ifp[ntake:]
        if ilo.size == 0 or not refine:
            break
    return max(best, 0.0), auc, passes


def _device_fns(engine, score, pose_xz, p_thresh, n_thresh, gt, row0, distinct=True, to_host=True):
    """(positive scores of the rectangle - as (distinct values, multiplicities), sorted and counted on the device, or
    as the raw float32 list for distinct=False (left on the device for to_host=False: the sharded job all-gathers it
    there) - and the counting function over its negatives)."""
    def count_fn(thresholds, rank):
        counts, bad, rank_sum = engine.pair_threshold_counts(score, thresholds, row0=row0, pose_xz=pose_xz, d_pos=p_thresh,
                                                             d_neg=n_thresh, gt=gt, rank=rank)
        if bad:
            raise ValueError("%d scores are negative or NaN" % bad)
        return counts, rank_sum
    pos, bad = engine.pair_positives(score, row0=row0, pose_xz=pose_xz, d_pos=p_thresh, d_neg=n_thresh, gt=gt)
    if bad:
        raise ValueError("%d scores are negative or NaN" % bad)
    if not distinct:
        return (pos.cpu().numpy() if to_host else pos), count_fn
    import torch
    u, mult = torch.unique(pos, sorted=True, return_counts=True)
    return (u.cpu().numpy(), mult.cpu().numpy().astype(np.int64)), count_fn


def pr_roc_device(engine, score, pose_xz=None, p_thresh=3.0, n_thresh=20.0, gt=None, row0=0, want_auc=True, refine=True):
    """(F1-max, ROC area, counting passes) of a score rectangle that stays on the device.  Ground truth from planar
    poses [M,2] (distance <= p_thresh positive, >= n_thresh negative, in between ignored) or explicit int8 labels
    (1 / 0 / -1)."""
    distinct, count_fn = _device_fns(engine, score, pose_xz, p_thresh, n_thresh, gt, row0)
    return pr_roc_from_counts(None, count_fn, want_auc=want_auc, distinct=distinct, refine=refine)


def f1_max_device(engine, score, pose_xz=None, p_thresh=3.0, n_thresh=20.0, gt=None, row0=0, one_call=True):
    """F1-max of a score rectangle that stays on the device -> (f1_max, counting passes).
    one_call (default): the engine's single entry point (sgpr_f1_max: positives, thresholds, counting passes and the
    F1 reduction all on the device, one small copy at the end); rectangles it reports as too large for that path - more
    than 2^20 positive pairs, more than 4095 values to settle in the second pass - and one_call=False take the multi-call
    path (pr_roc_device), which handles any size.  Both are exact."""
    if one_call and hasattr(engine, "f1_max"):
        res = engine.f1_max(score, row0=row0, pose_xz=pose_xz, d_pos=p_thresh, d_neg=n_thresh, gt=gt)
        status = int(res[1])
        if status == 2:
            raise ValueError("scores of labelled pairs are negative or NaN")
        if status == 0:
            return float(res[0]), int(res[4])
    f1, _, passes = pr_roc_device(engine, score, pose_xz, p_thresh, n_thresh, gt, row0, want_auc=False)
    return f1, passes


def roc_auc_device(engine, score, pose_xz=None, p_thresh=3.0, n_thresh=20.0, gt=None, row0=0):
    """ROC area (eval_batch.py:48-49) of a score rectangle that stays on the device: exact, one counting pass."""
    return pr_roc_device(engine, score, pose_xz, p_thresh, n_thresh, gt, row0, refine=False)[1]


def counts_of(score, gt):
    """numpy stand-ins for sgpr_pair_positives / sgpr_pair_threshold_counts (tests, small inputs): gt 1 / 0 / negative
    = ignored.  Returns (positive scores, count_fn)."""
    sc = np.ascontiguousarray(score, dtype=np.float32).ravel()
    cls = np.asarray(gt).ravel().astype(np.int64)
    pos, negs = sc[cls > 0], np.sort(sc[cls == 0])

    def count_fn(thresholds, rank):
        thr = np.asarray(thresholds, dtype=np.float32)
        b = np.searchsorted(thr, negs, side="right")                    # thresholds <= score
        counts = np.bincount(b, minlength=thr.size + 1).astype(np.int64)
        rank_sum = None
        if rank is not None:
            u, _, above = rank
            le = np.searchsorted(u, negs, side="right")
            gt_s = above[le]
            eq_s = np.where((le > 0) & (u[np.maximum(le, 1) - 1] == negs), above[np.maximum(le, 1) - 1] - gt_s, 0)
            rank_sum = int((2 * gt_s + eq_s).sum())
        return counts, rank_sum
    return pos, count_fn
