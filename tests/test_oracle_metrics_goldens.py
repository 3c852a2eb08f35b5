"""CPU: the metrics oracle (oracle/restate_metrics.py) against numbers produced by the reference's own functions
(oracle/make_goldens_metrics.py -> tests/golden/metrics_*.npz), plus hand-computable known answers."""
import glob
import json
import os

import numpy as np
import pytest

from multipitch_architectures_amd.synth import synth_eval_pair
from oracle import restate_metrics as RM

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "metrics_*.npz")))


def test_fixture_inventory():
    assert len(FILES) == 6


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[8:-4] for f in FILES])
@pytest.mark.parametrize("use_sklearn", [True, False])
def test_metrics_oracle_matches_reference(path, use_sklearn):
    g = np.load(path)
    targ, pred = synth_eval_pair(**json.loads(str(g["kwargs"])))
    got = RM.all_measures(targ, pred, threshold=float(g["threshold"]), use_sklearn=use_sklearn)
    assert list(g["measures"]) == RM.MEASURES
    for name, want in zip(RM.MEASURES, g["values"]):
        assert got[name] == pytest.approx(want, rel=1e-12, abs=1e-15), name


def test_known_answers():
    targ = np.array([[1, 0, 0, 1], [0, 0, 0, 0], [0, 1, 0, 0]], dtype=np.float64)
    pred = np.array([[0.9, 0.6, 0.1, 0.2], [0.1, 0.1, 0.1, 0.1], [0.2, 0.8, 0.7, 0.1]], dtype=np.float64)
    m = RM.all_measures(targ, pred, threshold=0.5, use_sklearn=False)
    # TP = 2 (0.9, 0.8), FP = 2 (0.6, 0.7), FN = 1 (0.2)
    assert m["precision"] == pytest.approx(0.5) and m["recall"] == pytest.approx(2 / 3)
    assert m["f_measure"] == pytest.approx(2 * 0.5 * (2 / 3) / (0.5 + 2 / 3))
    assert m["binary_accuracy"] == pytest.approx(9 / 12)
    # silent frame: both rows become/are proportional to the constant vector -> cosine 1
    c0 = (0.9 + 0.2) / np.sqrt(2) / np.sqrt(0.81 + 0.36 + 0.01 + 0.04)
    c2 = 0.8 / np.sqrt(0.04 + 0.64 + 0.49 + 0.01)
    assert m["cosine_sim"] == pytest.approx((c0 + 1.0 + c2) / 3)
    # ranking: positives 0.9, 0.8, 0.2; negatives above 0.2: 0.7, 0.6 ; ties at 0.1/0.2 with negatives
    # AUC by pair counting: P=3, N=9; 0.9 and 0.8 beat all 9; 0.2 beats the six 0.1s, ties one 0.2, loses to 0.7, 0.6
    assert m["roc_auc_measure"] == pytest.approx((9 + 9 + 6 + 0.5) / 27)
    assert m["average_precision_score"] == pytest.approx((1 / 3) * 1 + (1 / 3) * 1 + (1 / 3) * (3 / 6))
