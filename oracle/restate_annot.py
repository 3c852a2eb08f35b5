"""ORACLE (test infrastructure only -- never imported by the product path): CPU restatement of the reference's
note-list -> piano-roll conversion, ``compute_annotation_array_nooverlap`` (libdl/data_preprocessing/hcqt.py:205-272), and
of ``compute_hopsize_cqt`` (:9-27).  Pinned by tests/golden/annot_*.npz, which oracle/make_goldens_annot.py produces by
running the reference's own function definitions (tests/test_oracle_annot.py)."""
import numpy as np

HEIGHTS = {"pitch_class": 12, "pitch": 128, "instruments": 1}      # hcqt.py:219-226


def hopsize_cqt(fs_cqt_target, fs=22050, num_octaves=7):
    """hcqt.py:9-27: the CQT hop must be a multiple of 2^(octaves-1)"""
    factor = 2 ** (num_octaves - 1)
    n = np.round((fs / fs_cqt_target) / factor)
    hop = int(max(1, factor * n))
    return hop, fs / hop


def frame_indices(note_events, fs_hcqt, shorten=1.0):
    """hcqt.py:232-257: start / end frame of every event after the corrections that keep every note at least one frame
    long; raises AssertionError where the reference's own assertion fires"""
    ev = np.asarray(note_events, dtype=np.float64)
    t0, t1 = ev[:, 0].copy(), ev[:, 1].copy()
    if shorten != 1.0:
        t1 = t0 + shorten * (t1 - t0)                                   # :232-233
    s = np.floor(t0 * fs_hcqt).astype(np.int64)                         # :236
    e = np.floor(t1 * fs_hcqt).astype(np.int64)
    gone = (e - s) < 1                                                  # :239-240 events that vanish at this frame rate
    for v in np.unique(e[gone]):                                        # :243-247, ascending; later values see earlier shifts
        s = np.where(s == v, s + 1, s)
        e = np.where(e == v, e + 1, e)
    s = np.where(gone, s - 1, s)                                        # :249
    s = np.where((e - s) < 1, s - 1, s)                                 # :250-252
    if ((e - s) < 1).any():                                             # :255-257
        raise AssertionError("still events of length<1 after correction!")
    return s, e


def annotation_array_nooverlap(note_events, n_frames, fs_hcqt, annot_type="pitch_class", shorten=1.0):
    if annot_type not in HEIGHTS:
        raise AssertionError(["annotation type " + str(annot_type) + " not valid!"])
    H = HEIGHTS[annot_type]
    out = np.zeros((H, n_frames))
    s, e = frame_indices(note_events, fs_hcqt, shorten)
    ev = np.asarray(note_events, dtype=np.float64)
    for k in range(ev.shape[0]):
        p = ev[k, 2]
        row = int(np.mod(p, 12)) if annot_type == "pitch_class" else (int(p) if annot_type == "pitch" else 0)   # :263-268
        if row >= H or row < -H:
            raise IndexError(f"index {row} is out of bounds for axis 0 with size {H}")
        a, b, _ = slice(int(s[k]), int(e[k])).indices(n_frames)         # numpy slice semantics of :270 (negative starts wrap)
        out[row, a:b] = 1
    return out
