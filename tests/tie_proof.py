"""Proof that a graph whose embedding differs from the oracle's differs ONLY through near-tied kNN candidates.
TEST INFRASTRUCTURE (imports the oracle): used by tests/ and tools/seq_parity.py, never by the product.

dgcnn.knn (reference dgcnn.py:14-20) ranks candidate j for row i by pd[i][j] = -|x_i|^2 + 2 x_i.x_j - |x_j|^2 evaluated
in fp32 by expansion (a matmul plus two broadcast adds) and takes torch.topk.  Two candidates whose true squared
distances to x_i differ by less than the rounding error of that expansion are ordered by the rounding of the BLAS that
ran the matmul, not by the model: another correct implementation may pick the other one, after which the layer's max
over neighbours - and everything downstream - legitimately differs.  `prove_ties` turns "the embedding differs" into a
checkable statement:

  1. up to the first layer (per branch) whose neighbour sets differ, the layer inputs of both implementations agree to
     rounding (gate `input_tol`), and
  2. in that layer every row whose neighbour set differs swaps candidates a (ours) and b (the oracle's) whose squared
     distances to the row, recomputed in float64 from the ORACLE's own fp32 layer input, are closer than

         bound = fp32 + min(perturbation, fp32),   fp32 = c * 2^-24 * (|x_i|^2 + max(|x_a|^2, |x_b|^2)),

     fp32 being the fp32 expansion's error scale (the three terms of pd are each rounded at the magnitude of
     |x|^2; c = TIE_C, far below the worst-case (C + 2) of a C-term dot product) and `perturbation` the exact first-order
     effect of the measured rounding-level difference between the two implementations' inputs to that layer
     (2 |x_i - x_j| |dx_i - dx_j| + |dx_i - dx_j|^2 for j = a, b: Cauchy-Schwarz) - CAPPED at the fp32 term, so that the
     implementation under test cannot buy itself a wider bound with its own numerical error, and
  3. the REFERENCE's own arithmetic saw a tie, too: the oracle's fp32 ranking keys of a and b (dgcnn.py:15-17, its own
     precision) differ by at most REF_GAP_ULPS (6; 2 in the 12-channel first semantic layer) units in the last place of
     |x_i|^2 + max(|x_a|^2, |x_b|^2); in the coordinate layer, whose keys the kernel restates operation for operation, they must be EXACTLY equal (only
     torch.topk's order among equal keys can then differ from the kernel's lowest-index rule).
Layers after the first differing one are not examined: their inputs differ for a proven reason.

A selection bug (a wrong neighbour that is NOT a near-tie) fails 2 and 3; a numerical bug upstream fails 1.
"""
import numpy as np
import torch

TIE_C = 4.0            # multiples of 2^-24 (|x_i|^2 + |x_j|^2) a gap may have and still count as an fp32 tie
INPUT_TOL = 5e-6       # max |difference| of a layer input between the two implementations before the first flip (= the GPU
                       # tests' FEAT_TOL on layer outputs; the kernels deliver <= 7.2e-7)
# units in the last place of S = |x_i|^2 + |x_j|^2 the reference's own fp32 keys of a flip may be apart.  A key is
# (-|x_j|^2 - inner) - |x_i|^2 with inner = -2 x_i.x_j out of a 64-term fp32 matmul: three roundings at magnitude S plus
# the accumulation error of the dot product and of |x_j|^2 - about 3 ulp(S) per key, so two keys of one row whose true
# values coincide can sit up to ~6 ulp(S) apart in a correct fp32 evaluation, in either order (another BLAS, another
# summation order).  6 is the gate; the largest value observed is printed with every census (4.0: config 5, node_num 256).
# 2 ulp - tighter than the float64 bound TIE_C * 2^-24 * S = 2 .. 4 ulp(S) itself - refused one genuine fp32-level tie there.
REF_GAP_ULPS = 6.0
REF_GAP_ULPS_SEM1 = 2.0   # ... of the first semantic layer (12 input channels: 12-term dot products, one-hot rows are exact)
LAYERS = ["xyz1", "xyz2", "xyz3", "sem1", "sem2", "sem3"]


def oracle_trace(oracle, sd, feats, k):
    """feats [1, 3+L, N] (torch, any float dtype) -> (inputs [6] of [N, C] numpy, knn [6] of [N, k] numpy int64):
    every EdgeConv layer's input and the oracle's neighbour lists there (SG.dgcnn_conv_pass, sg_net.py:79-110).
    The inputs list carries, as attribute `.pd`, the oracle's own ranking keys pd [N, N] per layer (dgcnn.py:15-17 in its
    own precision): how far apart the REFERENCE's arithmetic saw two swapped candidates."""
    with torch.no_grad():
        _, lay = oracle.conv_pass(sd, feats, k, want_layers=True)
        ins = [feats[:, :3, :], lay["xyz1"], lay["xyz2"], feats[:, 3:, :], lay["sem1"], lay["sem2"]]
        knn = [oracle.knn(x, k)[0].numpy().astype(np.int64) for x in ins]
        pd = [oracle.neg_sq_dist(x)[0].numpy() for x in ins]
    out = _Trace(x[0].T.contiguous().numpy() for x in ins)
    out.pd = pd
    return out, knn


class _Trace(list):
    pd = None


def _canon(x):
    """index of the first row identical to row j: ties between feature-identical nodes (padding) are no ties at all"""
    _, first, inv = np.unique(x, axis=0, return_index=True, return_inverse=True)
    return first[np.asarray(inv).reshape(-1)]


def _multiset_diff(a, b):
    """elements of sorted int array a not matched in b (with multiplicity)"""
    out, b = [], list(b)
    for v in a:
        if v in b:
            b.remove(v)
        else:
            out.append(int(v))
    return out


def prove_ties(x_o, knn_o, x_h, knn_h, tie_c=TIE_C, input_tol=INPUT_TOL, ref_gap_ulps=REF_GAP_ULPS):
    """x_o / x_h: the six layer inputs [N, C] of the oracle / of the implementation under test; knn_o / knn_h: their
    neighbour lists [N, k].  Returns a report dict: `proven` (bool), `flips` (one entry per differing row of the first
    differing layer of each branch: layer, row, ours, theirs, gap, bound, ratio), `reason` when not proven."""
    rep = {"proven": True, "flips": [], "reason": "", "first_layer": {}}
    for branch in (0, 3):
        for li in range(branch, branch + 3):
            xo = np.asarray(x_o[li], dtype=np.float64)
            xh = np.asarray(x_h[li], dtype=np.float64)
            din = np.abs(xo - xh).max()
            if din > input_tol:
                rep["proven"] = False
                rep["reason"] += "%s: inputs differ by %.3g before any neighbour set does; " % (LAYERS[li], din)
                break
            canon = _canon(np.asarray(x_o[li]))
            mine = np.sort(canon[np.asarray(knn_h[li], dtype=np.int64)], -1)
            theirs = np.sort(canon[np.asarray(knn_o[li], dtype=np.int64)], -1)
            rows = np.flatnonzero((mine != theirs).any(-1))
            if rows.size == 0:
                continue
            rep["first_layer"]["xyz" if branch == 0 else "sem"] = LAYERS[li]
            nrm = (xo * xo).sum(1)
            for i in rows:
                a_list = _multiset_diff(mine[i], theirs[i])
                b_list = _multiset_diff(theirs[i], mine[i])
                if len(a_list) != len(b_list) or not a_list:
                    rep["proven"] = False
                    rep["reason"] += "%s row %d: neighbour multisets of different size; " % (LAYERS[li], i)
                    continue
                for a in a_list:
                    for b in b_list:
                        d2 = lambda j: float(((xo[i] - xo[j]) ** 2).sum())
                        gap = abs(d2(a) - d2(b))
                        fp32 = tie_c * 2.0 ** -24 * (nrm[i] + max(nrm[a], nrm[b]))
                        pert = 0.0
                        for j in (a, b):
                            dd = np.linalg.norm((xh[i] - xo[i]) - (xh[j] - xo[j]))
                            pert += 2.0 * np.sqrt(d2(j)) * dd + dd * dd
                        bound = fp32 + min(pert, fp32)
                        pd = getattr(x_o, "pd", None)
                        ref_gap = float(abs(np.float64(pd[li][i, a]) - np.float64(pd[li][i, b]))) if pd is not None else None
                        ulp = float(np.spacing(np.float32(nrm[i] + max(nrm[a], nrm[b]))))
                        rep["flips"].append({"layer": LAYERS[li], "row": int(i), "ours": a, "theirs": b, "gap": gap,
                                             "d2": d2(b), "fp32_bound": fp32, "perturbation": pert,
                                             "ratio": gap / bound, "ratio_fp32": gap / fp32, "ref_gap": ref_gap,
                                             "ref_gap_ulps": None if ref_gap is None else ref_gap / ulp})
                        if not gap <= bound:
                            rep["proven"] = False
                            rep["reason"] += ("%s row %d: candidates %d / %d are %.3g apart in d^2 (bound %.3g): not a "
                                              "tie; " % (LAYERS[li], i, a, b, gap, bound))
                        if ref_gap is not None:
                            # the reference's own keys: a tie in ITS arithmetic (exactly equal in the coordinate layer)
                            # (the 12-channel first semantic layer: a 12-term dot product leaves about an ulp per key)
                            allowed = 0.0 if li == 0 else (min(ref_gap_ulps, REF_GAP_ULPS_SEM1) if li == 3 else ref_gap_ulps) * ulp
                            if not ref_gap <= allowed:
                                rep["proven"] = False
                                rep["reason"] += ("%s row %d: the reference's fp32 keys of candidates %d / %d are %.3g apart "
                                                  "(%.2f ulp; allowed %.3g): not a tie in its arithmetic; "
                                                  % (LAYERS[li], i, a, b, ref_gap, ref_gap / ulp, allowed))
            break            # later layers of this branch differ for a proven (or disproven) reason
    return rep


def rows_at_risk(x_o, k, tie_c=TIE_C):
    """How common is the situation the proof accepts?  For the six oracle layer inputs: rows whose k-th and (k+1)-th
    DISTINCT candidates lie within the fp32 bound of each other -> (rows at risk, rows) per layer."""
    out = []
    for x in x_o:
        xo = np.asarray(x, dtype=np.float64)
        canon = _canon(np.asarray(x))
        nrm = (xo * xo).sum(1)
        d2 = ((xo[:, None, :] - xo[None, :, :]) ** 2).sum(-1)
        order = np.argsort(d2, axis=1, kind="stable")
        risk = 0
        for i in range(xo.shape[0]):
            a, b = order[i, k - 1], order[i, k] if xo.shape[0] > k else order[i, k - 1]
            if canon[a] == canon[b]:
                continue
            if abs(d2[i, a] - d2[i, b]) <= tie_c * 2.0 ** -24 * (nrm[i] + max(nrm[a], nrm[b])):
                risk += 1
        out.append((risk, xo.shape[0]))
    return out


def hip_trace(eng, centers, labels, k):
    """One graph through sgpr_embed_debug -> (inputs [6] of [N, C], knn [6] of [N, k], pooled [32]) in the layout of
    oracle_trace: the kernel's own layer outputs are the next layer's inputs."""
    pooled, _, _, layers, knn = eng.embed(centers[None], labels[None], k, debug=True)
    layers, knn = layers[0].cpu().numpy(), knn[0].cpu().numpy().astype(np.int64)
    n = labels.shape[0]
    onehot = np.zeros((n, 12), dtype=np.float32)
    real = labels >= 0
    onehot[np.flatnonzero(real), labels[real]] = 1.0
    ins = [np.asarray(centers, dtype=np.float32), layers[0], layers[1], onehot, layers[3], layers[4]]
    return ins, [knn[i] for i in range(6)], pooled[0].cpu().numpy()


def prove_graph(eng, oracle, sd, centers_g, labels_g, k, pooled_g=None):
    """One graph whose embedding (or one of whose scores) differs from the oracle's: the report of prove_ties, with
    `proven` additionally requiring that a flip was found and that the debug instance reproduces the production launch's
    pooled vector (pooled_g, when given)."""
    from sg_pr_amd import synth
    x_h, knn_h, p_dbg = hip_trace(eng, centers_g, labels_g, k)
    x_o, knn_o = oracle_trace(oracle, sd, torch.from_numpy(synth.dense_features(centers_g[None], labels_g[None])), k)
    rep = prove_ties(x_o, knn_o, x_h, knn_h)
    if pooled_g is not None and not np.array_equal(p_dbg, np.asarray(pooled_g)):
        rep["proven"] = False
        rep["reason"] += "the debug instance's pooled vector is not the production launch's; "
    if not rep["flips"]:
        rep["proven"] = False
        rep["reason"] += "no neighbour set differs; "
    return rep


def census(eng, oracle, sd, centers, labels, poses, k=10, flag_tol=2e-4, score_tol=1e-4, log=print, oracle_rows=None):
    """Whole-sequence parity of the HIP path against the oracle with every deviation accounted for.  Returns a dict of
    the figures the test gates; `log` receives the human-readable lines (profiles/r04_seq_parity.txt).
    oracle_rows: score only these rows of the matrix with the oracle (None = the full square)."""
    import time
    from sg_pr_amd import synth
    G = labels.shape[0]
    t0 = time.time()
    ref = []
    for s in range(0, G, 256):
        ref.append(oracle.embed(sd, torch.from_numpy(synth.dense_features(centers[s:s + 256], labels[s:s + 256])), k)[0])
    ref = torch.cat(ref)
    log("oracle: %d graphs embedded in %.1f s (%d threads)" % (G, time.time() - t0, torch.get_num_threads()))
    pooled = eng.embed(centers, labels, k)[0]
    dev = (pooled.cpu() - ref).abs().amax(1).numpy()
    log("max|d pooled| per graph, percentiles 50 / 99 / 99.9 / max: %.2e %.2e %.2e %.2e"
        % tuple(np.quantile(dev, [0.5, 0.99, 0.999, 1.0])))
    flagged = np.flatnonzero(dev > flag_tol)
    log("graphs with |d pooled| > %.0e: %d of %d %s" % (flag_tol, flagged.size, G, flagged.tolist()))
    out = {"graphs": G, "flagged": flagged, "proven": [], "unproven": [], "ratios": []}
    for g in flagged:
        x_h, knn_h, p_dbg = hip_trace(eng, centers[g], labels[g], k)
        same = bool(np.array_equal(p_dbg, pooled[g].cpu().numpy()))
        x_o, knn_o = oracle_trace(oracle, sd, torch.from_numpy(synth.dense_features(centers[g:g + 1], labels[g:g + 1])), k)
        rep = prove_ties(x_o, knn_o, x_h, knn_h)
        if not same:
            rep["proven"] = False
            rep["reason"] += "the debug instance's pooled vector is not the production launch's; "
        if not rep["flips"]:
            rep["proven"] = False
            rep["reason"] += "pooled vectors differ by %.3g but no neighbour set does; " % dev[g]
        (out["proven"] if rep["proven"] else out["unproven"]).append(int(g))
        for f in rep["flips"]:
            out["ratios"].append(f["ratio"])
            log("  graph %d (|d pooled| %.2e) %s row %d: ours %d / oracle's %d, d^2 = %.6g, gap %.3g = %.3f of the fp32 "
                "bound %.3g alone (input rounding term %.3g, capped at the fp32 term: %.2f of the bound); the oracle's own "
                "fp32 keys of the two: %s"
                % (g, dev[g], f["layer"], f["row"], f["ours"], f["theirs"], f["d2"], f["gap"], f["ratio_fp32"],
                   f["fp32_bound"], f["perturbation"], f["ratio"],
                   "EQUAL (torch.topk's tie order decides)" if f["ref_gap"] == 0.0
                   else "%.3g apart = %.2f ulp of |x_i|^2 + |x_j|^2" % (f["ref_gap"], f["ref_gap_ulps"])))
        if not rep["proven"]:
            log("  graph %d NOT PROVEN: %s" % (g, rep["reason"]))
        out.setdefault("ratios_fp32", []).extend(f["ratio_fp32"] for f in rep["flips"])
        out.setdefault("ref_gap_ulps", []).extend(f["ref_gap_ulps"] for f in rep["flips"] if f["ref_gap_ulps"] is not None)
    log("proven ties: %d of %d flagged graphs; largest gap / bound %.3f, largest gap / fp32 term alone %.3f, the reference's "
        "own keys at most %.2f ulp apart (input gate %.0e, reference-key gate %.1f ulp)"
        % (len(out["proven"]), flagged.size, max(out["ratios"]) if out["ratios"] else 0.0,
           max(out.get("ratios_fp32") or [0.0]), max(out.get("ref_gap_ulps") or [0.0]), INPUT_TOL, REF_GAP_ULPS))
    clean = np.setdiff1d(np.arange(G), flagged)
    out["clean_pooled_max"] = float(dev[clean].max()) if clean.size else 0.0
    # ---- score matrices
    rows = np.arange(G) if oracle_rows is None else np.asarray(oracle_rows)
    t0 = time.time()
    s_o = oracle.score_all_pairs(sd, ref[rows], ref, chunk=32).numpy()
    log("oracle: %d x %d scores in %.1f s" % (rows.size, G, time.time() - t0))
    s_h = eng.score_all_pairs(pooled[torch.from_numpy(rows).to(pooled.device)], pooled).cpu().numpy()
    d = np.abs(s_h - s_o)
    touch = np.zeros(G, dtype=bool)
    touch[flagged] = True
    tm = touch[rows][:, None] | touch[None, :]
    out["scores"] = int(d.size)
    out["scores_off"] = int((d > score_tol).sum())
    out["scores_off_clean"] = int((d[~tm] > score_tol).sum())
    out["score_max_clean"] = float(d[~tm].max()) if (~tm).any() else 0.0
    out["score_max"] = float(d.max())
    log("scores: %d; |d| > %.0e: %d (%.4f %%), all of them in rows / columns of the %d flagged graphs: %s; max |d| %.3e, "
        "max |d| between clean graphs %.3e" % (d.size, score_tol, out["scores_off"], 100.0 * out["scores_off"] / d.size,
                                                flagged.size, out["scores_off_clean"] == 0, out["score_max"],
                                                out["score_max_clean"]))
    # ---- F1-max (eval_batch.py:69-87) of both matrices, ground truth from the poses
    xz = poses[:, [3, 11]]
    dist = np.sqrt(((xz[rows][:, None, :] - xz[None, :, :]) ** 2).sum(-1))
    lab = np.where(dist <= 3.0, 1.0, np.where(dist >= 20.0, 0.0, -1.0))
    m = lab >= 0
    out["positives"], out["negatives"] = int((lab == 1).sum()), int((lab == 0).sum())
    out["f1_oracle"] = oracle.f1_max(lab[m], s_o[m])
    out["f1_hip"] = oracle.f1_max(lab[m], s_h[m])
    log("F1-max over %d positive / %d negative pairs: oracle matrix %.9f, HIP matrix %.9f, |d| %.3e"
        % (out["positives"], out["negatives"], out["f1_oracle"], out["f1_hip"], abs(out["f1_oracle"] - out["f1_hip"])))
    mc = m & ~tm                                          # ... and over the pairs between graphs whose embeddings agree
    out["f1_oracle_clean"] = oracle.f1_max(lab[mc], s_o[mc])
    out["f1_hip_clean"] = oracle.f1_max(lab[mc], s_h[mc])
    log("F1-max without the rows / columns of the flagged graphs: oracle %.9f, HIP %.9f, |d| %.3e"
        % (out["f1_oracle_clean"], out["f1_hip_clean"], abs(out["f1_oracle_clean"] - out["f1_hip_clean"])))
    out["pooled"], out["ref"], out["s_h"] = pooled, ref, s_h
    return out
