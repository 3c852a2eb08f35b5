"""SURVEY 8 f4 (the pinnable half): note list -> piano roll on the GPU (csrc/annot.hip through the C ABI) against the
fixtures the reference's own ``compute_annotation_array_nooverlap`` (libdl/data_preprocessing/hcqt.py:205-272) produced on the
note list it ships (data/MusicNet/csv/2382_...csv) and on lists that force its correction branches.  Integer / index work:
the arrays must be identical; the reference's assertion must fire where it fires upstream."""
import glob
import os

import numpy as np
import pytest
import torch

from multipitch_architectures_amd.data_preprocessing import (annotation_array_nooverlap_device,
                                                             compute_annotation_array_nooverlap, compute_hopsize_cqt)
from oracle import restate_annot as RA

pytestmark = pytest.mark.gpu
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN_DIR, "annot_*.npz")))


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("path", CASES, ids=lambda p: os.path.basename(p)[6:-4])
def test_piano_roll_is_identical_to_the_reference(dev, path):
    g = np.load(path)
    ev, n_frames, fs, kind, shorten = g["events"].copy(), int(g["n_frames"]), float(g["fs"]), str(g["kind"]), float(g["shorten"])
    f_hcqt = np.zeros((1, n_frames, 1))
    if "bits" not in g.files:
        with pytest.raises(AssertionError, match="still events of length<1"):
            compute_annotation_array_nooverlap(ev, f_hcqt, fs, annot_type=kind, shorten=shorten)
        return
    shape = tuple(int(v) for v in g["shape"])
    want = np.unpackbits(g["bits"])[: shape[0] * shape[1]].reshape(shape).astype(np.float64)
    before = ev.copy()
    got = compute_annotation_array_nooverlap(ev, f_hcqt, fs, annot_type=kind, shorten=shorten)
    assert isinstance(got, np.ndarray) and got.dtype == np.float64 and got.shape == want.shape
    assert np.array_equal(got, want)
    if shorten != 1.0:       # the reference shortens the caller's array in place
        assert np.array_equal(ev[:, 1], before[:, 0] + shorten * (before[:, 1] - before[:, 0]))


def test_random_lists_against_the_oracle(dev):
    """other sizes / rates than the fixtures: empty list, one event, negative start frames (slice wrap), long lists"""
    rng = np.random.default_rng(5)
    for n, fs, kind, shorten in [(0, 43.0, "pitch", 1.0), (1, 43.0, "pitch_class", 1.0), (5000, 86.1328125, "pitch", 0.7),
                                 (300, 7.0, "instruments", 1.0), (2000, 43.066, "pitch_class", 0.9)]:
        st = np.sort(rng.uniform(-0.01, 120.0, n))
        ev = np.stack([st, st + rng.uniform(0.08, 3.0, n), rng.integers(0, 128, n).astype(np.float64), np.zeros(n)], 1) \
            if n else np.zeros((0, 4))
        n_frames = int(120 * fs) + 5
        try:
            want = RA.annotation_array_nooverlap(ev.copy(), n_frames, fs, kind, shorten)
        except AssertionError:
            with pytest.raises(AssertionError):
                annotation_array_nooverlap_device(ev, n_frames, fs, kind, shorten)
            continue
        got = annotation_array_nooverlap_device(ev, n_frames, fs, kind, shorten).cpu().numpy()
        assert np.array_equal(got, want), (n, fs, kind)


def test_errors_and_hopsize(dev):
    assert compute_hopsize_cqt(50, fs=22050, num_octaves=9) == RA.hopsize_cqt(50, fs=22050, num_octaves=9)
    assert compute_hopsize_cqt(91) == RA.hopsize_cqt(91)
    with pytest.raises(AssertionError):
        annotation_array_nooverlap_device(np.zeros((1, 4)) + [[0.0, 1.0, 60.0, 0.0]], 100, 43.0, "chroma")
    with pytest.raises(IndexError):
        annotation_array_nooverlap_device(np.array([[0.0, 1.0, 130.0, 0.0]]), 100, 43.0, "pitch")
