"""Pins oracle/restate_annot.py against tests/golden/annot_*.npz (produced by the reference's own function definitions,
oracle/make_goldens_annot.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import restate_annot as RA

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "annot_*.npz")))


def load(path):
    g = np.load(path)
    want = None
    if "bits" in g.files:
        shape = tuple(int(v) for v in g["shape"])
        want = np.unpackbits(g["bits"])[: shape[0] * shape[1]].reshape(shape).astype(np.float64)
    return g, want


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[6:-4])
def test_oracle_reproduces_the_reference_piano_rolls(path):
    g, want = load(path)
    args = (g["events"].copy(), int(g["n_frames"]), float(g["fs"]), str(g["kind"]), float(g["shorten"]))
    if want is None:
        with pytest.raises(AssertionError, match="still events of length<1"):
            RA.annotation_array_nooverlap(*args)
        return
    got = RA.annotation_array_nooverlap(*args)
    assert got.dtype == np.float64 and got.shape == want.shape
    assert np.array_equal(got, want)


def test_hopsize_matches_the_reference_values():
    g = np.load(os.path.join(GOLDEN_DIR, "annot_musicnet_pitch.npz"))
    hop, fs = RA.hopsize_cqt(50, fs=22050, num_octaves=9)
    assert hop == int(g["hopsize"]) == 512 and fs == float(g["fs"])
