"""GPU: evaluation measures (mpa_eval_measures through the C ABI) against the metrics oracle and the numbers the
reference's own functions produced (tests/golden/metrics_*.npz) -- SURVEY section 8 f2."""
import glob
import json
import os

import numpy as np
import pytest

from multipitch_architectures_amd.synth import synth_eval_pair

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "metrics_*.npz")))


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[8:-4] for f in FILES])
def test_measures_match_reference_goldens(path):
    from multipitch_architectures_amd.metrics import MEASURES, calculate_eval_measures
    g = np.load(path)
    targ, pred = synth_eval_pair(**json.loads(str(g["kwargs"])))
    got = calculate_eval_measures(targ, pred, list(g["measures"]), threshold=float(g["threshold"]))
    assert list(got) == MEASURES
    for name, want in zip(g["measures"], g["values"]):
        assert got[str(name)] == pytest.approx(want, rel=1e-11, abs=1e-14), name


@pytest.mark.parametrize("n,k,thr", [(1, 72, 0.5), (3, 12, 0.4), (257, 72, 0.4), (1000, 84, 0.3), (20000, 72, 0.4)])
def test_measures_match_oracle(n, k, thr):
    from multipitch_architectures_amd.metrics import calculate_eval_measures, MEASURES
    from oracle import restate_metrics as RM
    targ, pred = synth_eval_pair(n_frames=n, n_bins=k, seed=n + k, quant=200 if n % 2 else None)
    if n == 1:
        targ[0, 5] = 1.0; targ[0, 9] = 1.0
    want = RM.all_measures(targ, pred, threshold=thr, use_sklearn=False)
    got = calculate_eval_measures(targ, pred, MEASURES, threshold=thr)
    for m in MEASURES:
        assert got[m] == pytest.approx(want[m], rel=1e-11, abs=1e-14), m


def test_known_answers_and_single_measure():
    import torch
    from multipitch_architectures_amd.metrics import calculate_single_measure
    targ = np.array([[1, 0, 0, 1] * 3, [0, 0, 0, 0] * 3, [0, 1, 0, 0] * 3], dtype=np.float32)
    pred = np.array([[0.9, 0.6, 0.1, 0.2] * 3, [0.1, 0.1, 0.1, 0.1] * 3, [0.2, 0.8, 0.7, 0.1] * 3], dtype=np.float32)
    assert calculate_single_measure(targ, pred, "precision", threshold=0.5) == pytest.approx(0.5)
    assert calculate_single_measure(targ, pred, "recall", threshold=0.5) == pytest.approx(2 / 3)
    assert calculate_single_measure(targ, pred, "binary_accuracy", threshold=0.5) == pytest.approx(9 / 12)
    # device tensors are accepted as they are (predictions stay on the GPU)
    f = calculate_single_measure(torch.from_numpy(targ).cuda(), torch.from_numpy(pred).cuda(), "f_measure", threshold=0.5)
    assert f == pytest.approx(2 * 0.5 * (2 / 3) / (0.5 + 2 / 3))
    with pytest.raises(AssertionError):
        calculate_single_measure(targ, pred, "no_such_measure")
    with pytest.raises(ValueError):
        calculate_single_measure(np.zeros_like(targ), pred, "roc_auc_measure")


def test_threshold_compare_is_done_in_double():
    # the scripts compare float64(pred) >= 0.4; 0.4f rounds *up*, so a float32 prediction equal to 0.4f counts as positive
    from multipitch_architectures_amd.metrics.eval_metrics import raw_measures
    targ = np.ones((1, 12), dtype=np.float32)
    pred = np.full((1, 12), np.float32(0.4), dtype=np.float32)
    pred[0, :6] = np.nextafter(np.float32(0.4), np.float32(0))
    raw = raw_measures(targ, pred, threshold=0.4)
    assert raw[11] == 6 and raw[13] == 6


def test_aggregate_files():
    from multipitch_architectures_amd.metrics.eval_metrics import aggregate_files
    mean, fw = aggregate_files([[1.0, 0.0], [0.0, 1.0]], [1000, 3000])
    assert list(mean) == [0.5, 0.5] and list(fw) == [0.25, 0.75]
