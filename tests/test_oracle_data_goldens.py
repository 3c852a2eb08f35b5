"""CPU: the data oracle (oracle/restate_data.py) against vectors produced by the imported reference
(oracle/make_goldens_data.py -> tests/golden/data_*.npz)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from multipitch_architectures_amd.synth import synth_file
from oracle import restate_data as RD

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "data_*.npz")))


def load_case(path):
    g = np.load(path)
    params = json.loads(str(g["params"]))
    inputs, targets = synth_file(frames=400, n_bins_out=int(g["n_out"]), seed=77)
    return g, params, torch.from_numpy(inputs), torch.from_numpy(targets)


def test_fixture_inventory():
    assert len(FILES) == 8


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[5:-4] for f in FILES])
def test_data_oracle_matches_reference(path):
    g, params, inputs, targets = load_case(path)
    assert RD.dataset_len(inputs, params) == int(g["len"])
    seen_tune, seen_transp = set(), set()
    for k, (index, seed) in enumerate(g["items"]):
        torch.manual_seed(int(seed))
        X, y, d = RD.context_patch(inputs, targets, params, int(index))
        seen_tune.add(d["tune2"]); seen_transp.add(np.sign(d["transp"]))
        assert tuple(X.shape) == tuple(g[f"{k}.shape"])
        Xn = X.numpy()
        # np.log (reference, :106) vs torch.log differ by at most 1 ulp
        np.testing.assert_allclose(Xn.ravel()[::11], g[f"{k}.xs"], rtol=3e-7, atol=1e-9)
        st = g[f"{k}.stats"]
        np.testing.assert_allclose([Xn.astype(np.float64).sum(), np.abs(Xn).astype(np.float64).sum(), Xn.max()], st,
                                   rtol=1e-6)
        np.testing.assert_array_equal(y.numpy(), g[f"{k}.y"])
    if os.path.basename(path) == "data_train.npz":
        assert seen_tune == {-2, -1, 0, 1, 2} and seen_transp == {-1, 0, 1}


def test_eq_offsets():
    # hcqt_datasets.py:90-93 for the harmonics [0.5, 1, 2, 3, 4, 5]
    assert [RD.harmonic_offset(h) for h in range(6)] == [-36, 0, 36, 57, 72, 83]
