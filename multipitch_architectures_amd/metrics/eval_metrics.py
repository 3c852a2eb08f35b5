"""GPU replacement of ``libdl/metrics/eval_metrics.py`` ``calculate_single_measure`` (:8-116) and
``calculate_eval_measures`` (:120-155): same names, argument order and measure names; one call evaluates *all*
measures of a recording in a handful of HIP launches (``mpa_eval_measures``), so predictions never leave the GPU
between the network and the printed F-score (the scripts copy every batch to the host, exp126a...py:432-438).

Arithmetic is float64 on the device, like the reference's numpy code.  ``calculate_mpe_measures_mireval`` (:159-193,
mir_eval/librosa) is not built.  No CPU path: fails loudly without the HIP library and a GPU.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib as L

MEASURES = ["precision", "recall", "f_measure", "cosine_sim", "binary_crossentropy", "euclidean_distance",
            "binary_accuracy", "soft_accuracy", "accum_energy", "roc_auc_measure", "average_precision_score"]


def _dev(a, device):
    t = torch.as_tensor(a) if not isinstance(a, torch.Tensor) else a
    return t.detach().to(device, torch.float32).contiguous()


def raw_measures(targets, predictions, threshold=0.5, device="cuda:0"):
    """All measures of one (n_frames, n_bins) recording -> host numpy array of 16 doubles (see include/mpa.h)."""
    lib = L.load()
    if not torch.cuda.is_available():
        raise RuntimeError("the evaluation measures run on the GPU; there is no CPU path")
    targ, pred = _dev(targets, device), _dev(predictions, device)
    assert targ.shape == pred.shape, "Error: Targets and predictions have different shape!"      # eval_metrics.py:42
    assert targ.dim() == 2
    n, k = targ.shape
    if k % 12 != 0:                                                                                 # :44-46
        print("WARNING: Shape of input is " + str(tuple(targ.shape)) +
              ", expect features (bins) as second dimension. Please make sure that size is correct!")
    nbytes = lib.mpa_eval_measures_workspace(n, k)
    if nbytes < 0:
        L.check(int(nbytes), "mpa_eval_measures_workspace")
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=targ.device)
    out = torch.empty(16, dtype=torch.float64, device=targ.device)
    vp = lambda t: ctypes.c_void_p(t.data_ptr())
    with torch.cuda.device(targ.device):
        rc = lib.mpa_eval_measures(vp(targ), vp(pred), n, k, float(threshold), vp(out), vp(ws), int(nbytes),
                                   ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.check(rc, "mpa_eval_measures")
    return out.cpu().numpy()


def calculate_eval_measures(targets, predictions, measures, threshold=0.5, save_roc_plot=False, path_output="roc.pdf",
                            device="cuda:0"):
    """dict measure name -> value, in the order of ``measures`` (eval_metrics.py:147-153)."""
    for m in measures:
        assert m in MEASURES, "ERROR: Evaluation measure " + str(m) + " not implemented!"          # :112-113
    if save_roc_plot:
        raise NotImplementedError("ROC plots are not built")
    raw = raw_measures(targets, predictions, threshold, device)
    vals = dict(zip(MEASURES, (float(v) for v in raw[:11])))
    if "roc_auc_measure" in measures and (raw[14] == 0 or raw[15] == 0):
        # scikit-learn's behaviour, which the reference inherits (:90)
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    if raw[14] == 0:
        vals["average_precision_score"] = -0.0 if math.isnan(vals["average_precision_score"]) else vals["average_precision_score"]
    return {m: vals[m] for m in measures}


def calculate_single_measure(targets, predictions, measure, threshold=0.5, save_roc_plot=False, path_output="roc.pdf",
                             device="cuda:0"):
    return calculate_eval_measures(targets, predictions, [measure], threshold, save_roc_plot, path_output, device)[measure]


def aggregate_files(per_file, n_frames):
    """file-wise mean and frame-weighted mean of per-recording measure vectors (exp126a...py:457-460,470,490)."""
    v = np.asarray(per_file, dtype=np.float64)
    kf = np.asarray(n_frames, dtype=np.float64) / 1000.0
    return v.mean(axis=0), (v * kf[:, None]).sum(axis=0) / kf.sum()
