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
# Device-side F1-max: exact, sort-free.  The engine counts (negative, positive) pairs by radix bins of
# the score's fp32 key (sgpr_pair_histogram); this module walks the cumulative counts and asks for finer passes only
# on the bins that can still contain the maximum.  `hist_fn(prefix_bits, bits, prefixes)` -> uint64 [n, 2^bits, 2].
_LEVEL_BITS = (12, 12, 8)


def _f1(tp, fp, pos):
    tp = np.asarray(tp, dtype=np.float64)
    fp = np.asarray(fp, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = np.where(tp + fp > 0, tp / (tp + fp), 0.0)
        r = tp / pos if pos > 0 else np.ones_like(tp)
        f = 2 * p * r / (p + r)
    return np.nan_to_num(f)


def f1_max_from_histograms(hist_fn, max_prefixes=4, max_pending=256):
    """F1-max of eval_batch.py:85-87 from class-wise score histograms; returns (f1_max, passes).
    Raises RuntimeError if more than `max_pending` bins stay undecided (use a sorted path then)."""
    h = np.asarray(hist_fn(0, _LEVEL_BITS[0], (0,)), dtype=np.uint64)[0].astype(np.float64)
    pos = float(h[:, 1].sum())
    best = 0.0
    passes = 1
    # pending: (prefix value, prefix_bits, TP above the bin, FP above the bin, upper bound of F1 inside the bin)
    pending = []

    def walk(hb, prefix, pbits, tp0, fp0, last):
        nonlocal best
        # bins in descending key order: cumulative counts ABOVE each bin, then including it
        p_desc, n_desc = hb[::-1, 1], hb[::-1, 0]
        tp_above = tp0 + np.concatenate(([0.0], np.cumsum(p_desc)[:-1]))
        fp_above = fp0 + np.concatenate(([0.0], np.cumsum(n_desc)[:-1]))
        occupied = (p_desc + n_desc) > 0
        if not occupied.any():
            return
        edge = _f1(tp_above + p_desc, fp_above + n_desc, pos)          # threshold = smallest score of the bin
        best = max(best, float(edge[occupied].max()))
        if last:
            return
        ub = _f1(tp_above + p_desc, fp_above, pos)                     # all its positives, none of its negatives
        nb = hb.shape[0]
        lg = nb.bit_length() - 1
        for i in np.nonzero(occupied & (p_desc > 0) & (n_desc > 0) & (ub > best))[0]:
            b = nb - 1 - int(i)
            pending.append(((prefix << lg) | b, pbits + lg, tp_above[i], fp_above[i], ub[i]))

    walk(h, 0, 0, 0.0, 0.0, False)
    level = 1
    while pending and level < len(_LEVEL_BITS):
        todo = [c for c in pending if c[4] > best]
        pending = []
        if len(todo) > max_pending:
            raise RuntimeError("f1_max_from_histograms: %d undecided bins" % len(todo))
        bits = _LEVEL_BITS[level]
        last = level == len(_LEVEL_BITS) - 1
        todo.sort(key=lambda c: -c[4])                                  # most promising first: raises `best` early
        for s in range(0, len(todo), max_prefixes):
            group = [c for c in todo[s:s + max_prefixes] if c[4] > best]
            if not group:
                continue
            hg = np.asarray(hist_fn(group[0][1], bits, tuple(c[0] for c in group)), dtype=np.uint64).astype(np.float64)
            passes += 1
            for c, hb in zip(group, hg):
                walk(hb, c[0], c[1], c[2], c[3], last)
        level += 1
    return best, passes


def roc_auc_from_histograms(hist_fn, tol=1e-9, max_passes=64, max_prefixes=4):
    """Area under the ROC curve (eval_batch.py:48-49) from the same class-wise score histograms, no sort:
    AUC * P * N = #{(pos, neg): s_pos > s_neg} + 0.5 #{s_pos == s_neg}.  Pairs in different bins are decided by the
    bins; a bin holding p positives and n negatives leaves p * n pairs open, counted as ties (= the trapezoid rule)
    until the bin is refined - largest p * n first, down to single fp32 values, where ties are exact.
    Returns (auc, half_width): the true value lies within +- half_width (0 when every open bin was resolved or
    `tol` when the refinement stopped there)."""
    h = np.asarray(hist_fn(0, _LEVEL_BITS[0], (0,)), dtype=np.uint64)[0].astype(np.float64)
    pos, neg = float(h[:, 1].sum()), float(h[:, 0].sum())
    if pos <= 0 or neg <= 0:
        return float("nan"), 0.0
    known = 0.0          # decided pairs (positive above negative)
    open_bins = []       # [p * n, prefix, prefix_bits, level]

    def absorb(hb, prefix, pbits, level):
        nonlocal known
        p_b, n_b = hb[:, 1], hb[:, 0]
        below = np.concatenate(([0.0], np.cumsum(n_b)[:-1]))          # negatives in lower bins of this histogram
        known += float((p_b * below).sum())
        lg = hb.shape[0].bit_length() - 1
        last = pbits + lg >= 32
        for b in np.nonzero((p_b > 0) & (n_b > 0))[0]:
            if last:
                known += 0.5 * p_b[b] * n_b[b]                        # single fp32 value: a true tie
            else:
                open_bins.append([p_b[b] * n_b[b], (prefix << lg) | int(b), pbits + lg, level + 1])

    absorb(h, 0, 0, 0)
    passes = 1
    while open_bins and passes < max_passes:
        if 0.5 * sum(o[0] for o in open_bins) / (pos * neg) <= tol:
            break
        open_bins.sort(key=lambda o: -o[0])
        lvl_bits = open_bins[0][2]
        group = [o for o in open_bins if o[2] == lvl_bits][:max_prefixes]
        for o in group:
            open_bins.remove(o)
        bits = _LEVEL_BITS[group[0][3]]
        hg = np.asarray(hist_fn(lvl_bits, bits, tuple(o[1] for o in group)), dtype=np.uint64).astype(np.float64)
        passes += 1
        for o, hb in zip(group, hg):
            absorb(hb, o[1], o[2], o[3])
    rest = sum(o[0] for o in open_bins)
    return (known + 0.5 * rest) / (pos * neg), 0.5 * rest / (pos * neg)


def roc_auc_device(engine, score, pose_xz=None, p_thresh=3.0, n_thresh=20.0, gt=None, row0=0, tol=1e-6, max_passes=64):
    """ROC AUC of a score rectangle that stays on the device (see roc_auc_from_histograms)."""
    def hist_fn(prefix_bits, bits, prefixes):
        return engine.pair_histogram(score, row0=row0, pose_xz=pose_xz, d_pos=p_thresh, d_neg=n_thresh, gt=gt,
                                     prefixes=prefixes, prefix_bits=prefix_bits, bits=bits)[0]
    return roc_auc_from_histograms(hist_fn, tol=tol, max_passes=max_passes)


def f1_max_device(engine, score, pose_xz=None, p_thresh=3.0, n_thresh=20.0, gt=None, row0=0):
    """F1-max of a score rectangle that stays on the device.  Ground truth from planar poses [M,2] (distance <=
    p_thresh positive, >= n_thresh negative, in between ignored) or explicit int8 labels (1 / 0 / -1)."""
    def hist_fn(prefix_bits, bits, prefixes):
        h, bad = engine.pair_histogram(score, row0=row0, pose_xz=pose_xz, d_pos=p_thresh, d_neg=n_thresh, gt=gt,
                                       prefixes=prefixes, prefix_bits=prefix_bits, bits=bits)
        if bad:
            raise ValueError("%d scores are negative or NaN" % bad)
        return h
    return f1_max_from_histograms(hist_fn)


def histograms_of(score, gt):
    """numpy stand-in for sgpr_pair_histogram (tests, small inputs): gt 1 / 0 / negative = ignored."""
    key = np.ascontiguousarray(score, dtype=np.float32).ravel().view(np.uint32).astype(np.uint64)
    cls = np.asarray(gt).ravel().astype(np.int64)
    keep = cls >= 0
    key, cls = key[keep], (cls[keep] != 0).astype(np.int64)

    def hist_fn(prefix_bits, bits, prefixes):
        out = np.zeros((len(prefixes), 1 << bits, 2), dtype=np.uint64)
        pre = key >> np.uint64(32 - prefix_bits) if prefix_bits else np.zeros_like(key)
        b = (key >> np.uint64(32 - prefix_bits - bits)) & np.uint64((1 << bits) - 1)
        for i, p in enumerate(prefixes):
            sel = pre == np.uint64(p)
            np.add.at(out[i], (b[sel].astype(np.int64), cls[sel]), 1)
        return out
    return hist_fn
