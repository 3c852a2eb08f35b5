"""ORACLE -- test infrastructure only.  Never imported by the product path.

numpy (float64) restatement of the evaluation measures the experiment scripts print
(``libdl/metrics/eval_metrics.py:8-116`` ``calculate_single_measure`` with
``libfmp/c5/c5s2_chord_rec_template.py:238-261`` ``compute_eval_measures`` and
``libfmp/c3/c3s1_post_processing.py:31-91`` ``normalize_feature_sequence(norm='2')``), written vectorised.
The two ranking measures call the same third-party functions the reference calls (scikit-learn
``roc_auc_score`` / ``average_precision_score``; installed here, not vendored by the reference) and additionally
``ranking_measures`` restates their published definition (distinct-threshold ROC trapezoid; AP = sum (R_n - R_{n-1}) P_n)
so that the GPU kernel's tie handling can be checked without scikit-learn.

Pinned by tests/golden/metrics_*.npz, produced by executing the reference's own three functions
(oracle/make_goldens_metrics.py).  The mir_eval-based measures (eval_metrics.py:159-193) are out of scope: mir_eval
and librosa are absent third-party packages.
"""
import numpy as np

MEASURES = ["precision", "recall", "f_measure", "cosine_sim", "binary_crossentropy", "euclidean_distance",
            "binary_accuracy", "soft_accuracy", "accum_energy", "roc_auc_measure", "average_precision_score"]


def prf(targ, pred_thresh):
    """c5s2_chord_rec_template.py:250-261."""
    tp = float(np.sum(np.logical_and(targ, pred_thresh)))
    fp = float(np.sum(pred_thresh > 0)) - tp
    fn = float(np.sum(targ > 0)) - tp
    p = r = f = 0.0
    if tp > 0:
        p, r = tp / (tp + fp), tp / (tp + fn)
        f = 2 * p * r / (p + r)
    return p, r, f, tp, fp, fn


def unit_rows(x, threshold=1e-10):
    """rows of x scaled to unit L2 norm; rows with norm <= threshold become 1/sqrt(K) (c3s1_post_processing.py:60-68,
    applied to frames; eval_metrics.py:68-69 passes the transposed matrices)."""
    x = x.astype(np.float64)
    nrm = np.sqrt(np.sum(x ** 2, axis=1, keepdims=True))
    out = np.full_like(x, 1.0 / np.sqrt(x.shape[1]))
    ok = nrm[:, 0] > threshold
    out[ok] = x[ok] / nrm[ok]
    return out


def ranking_measures(targ, pred):
    """ROC-AUC (trapezoid over distinct thresholds, from (0,0)) and average precision, from their definitions."""
    y = targ.ravel().astype(np.float64)
    s = pred.ravel().astype(np.float64)
    order = np.argsort(-s, kind="stable")
    y, s = y[order], s[order]
    last = np.r_[np.nonzero(np.diff(s))[0], y.size - 1]          # last element of each group of tied scores
    tps = np.cumsum(y)[last]
    fps = (last + 1) - tps
    P, N = tps[-1], fps[-1]
    tp0, fp0 = np.r_[0.0, tps[:-1]], np.r_[0.0, fps[:-1]]
    auc = np.sum((fps - fp0) * (tps + tp0) / 2) / (P * N)
    ap = np.sum((tps - tp0) / P * tps / (tps + fps))
    return auc, ap


def all_measures(targ, pred, threshold=0.5, use_sklearn=True):
    """dict of the 11 measures of exp180d...py:150-151, evaluated like eval_metrics.py:44-113 (float64)."""
    targ = np.asarray(targ, dtype=np.float64)
    pred = np.asarray(pred, dtype=np.float64)
    assert targ.shape == pred.shape
    eps = np.finfo(float).eps
    pt = pred >= threshold
    p, r, f, _, _, _ = prf(targ, pt)
    out = {"precision": p, "recall": r, "f_measure": f}
    out["cosine_sim"] = float(np.sum(unit_rows(targ) * unit_rows(pred)) / targ.shape[0])
    out["binary_crossentropy"] = float(-np.mean(targ * np.log2(pred + eps) + (1 - targ) * np.log2(1 - pred + eps)))
    out["euclidean_distance"] = float(np.mean(np.sqrt(np.sum((targ - pred) ** 2, axis=1))))
    out["binary_accuracy"] = float(np.mean(pt == targ))
    out["soft_accuracy"] = float(np.mean(targ * pred + (1 - targ) * (1 - pred)))
    out["accum_energy"] = float(np.mean(np.sum(targ * pred, axis=1) / (np.sum(targ, axis=1) + eps)))
    if use_sklearn:
        from sklearn import metrics as sk_metrics
        out["roc_auc_measure"] = float(sk_metrics.roc_auc_score(targ.ravel(), pred.ravel()))
        out["average_precision_score"] = float(sk_metrics.average_precision_score(targ.ravel(), pred.ravel()))
    else:
        out["roc_auc_measure"], out["average_precision_score"] = (float(v) for v in ranking_measures(targ, pred))
    return out
