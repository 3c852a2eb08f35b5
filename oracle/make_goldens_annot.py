"""ORACLE tooling -- generates tests/golden/annot_*.npz with the reference's own note-list -> piano-roll code
(the pinnable half).  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_annot.py

``libdl/data_preprocessing/hcqt.py`` cannot be imported here (its header imports IPython, librosa, numba), but
``compute_annotation_array_nooverlap`` (hcqt.py:205-272) and ``compute_hopsize_cqt`` (:9-27, ``@jit`` dropped: numba only
compiles it) use numpy alone.  As for the evaluation measures (make_goldens_metrics.py) the generator compiles exactly
those two function definitions from the reference's source file *where it lies* and runs them on the note list the
reference ships (data/MusicNet/csv/2382_Beethoven_OP130_StringQuartet.csv, prepared as in 01_precompute_features.ipynb
cell 7) and on derived lists that force the correction branches (events that vanish at the frame rate, chains of
vanishing end times, shortened notes).  The fixtures hold data only: the note events, the arguments and the bit-packed
resulting arrays.
"""
import ast
import csv
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def reference_function(path, name, namespace, drop_decorators=False):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name)
    if drop_decorators:
        fn.decorator_list = []
    exec(compile(ast.Module(body=[fn], type_ignores=[]), os.path.join(REF, path), "exec"), namespace)
    return namespace[name]


def musicnet_events():
    rows = []
    with open(os.path.join(REF, "data", "MusicNet", "csv", "2382_Beethoven_OP130_StringQuartet.csv")) as f:
        rd = csv.reader(f)
        next(rd)
        for r in rd:
            rows.append((float(r[0]) / 44100.0, float(r[1]) / 44100.0, float(r[3]), 0.0))     # notebook cell 7
    return np.array(rows, dtype=np.float64)


def main():
    ns = {"np": np}
    hop = reference_function("libdl/data_preprocessing/hcqt.py", "compute_hopsize_cqt", dict(ns), True)
    annot = reference_function("libdl/data_preprocessing/hcqt.py", "compute_annotation_array_nooverlap", dict(ns))
    ev = musicnet_events()
    hopsize, fs_hcqt = hop(50, fs=22050, num_octaves=9)         # compute_efficient_hcqt's effective octave count, :113-114
    rng = np.random.default_rng(11)
    # a synthetic list that forces the corrections: very short notes, equal start / end times, chains of vanishing ends
    n = 600
    st = np.sort(rng.uniform(0.0, 30.0, n))
    du = np.where(rng.random(n) < 0.5, rng.uniform(0.0, 0.03, n), rng.uniform(0.03, 1.5, n))
    dense = np.stack([st, st + du, rng.integers(21, 109, n).astype(np.float64), np.zeros(n)], 1)
    dense[::7, 1] = dense[::7, 0]                               # zero-length events
    dense[1::9, 0] = dense[0:-1:9, 1][: dense[1::9].shape[0]]    # a note that starts exactly where another one ends
    # a sparser one whose corrections succeed (isolated short notes)
    n2 = 400
    st2 = np.sort(rng.uniform(0.0, 60.0, n2))
    du2 = np.where(rng.random(n2) < 0.15, rng.uniform(0.0, 0.02, n2), rng.uniform(0.05, 2.0, n2))
    sparse = np.stack([st2, st2 + du2, rng.integers(21, 109, n2).astype(np.float64), np.zeros(n2)], 1)
    sparse[::41, 1] = sparse[::41, 0]
    cases = {
        "musicnet_pitch": (ev, fs_hcqt, "pitch", 1.0),
        "musicnet_pitchclass": (ev, fs_hcqt, "pitch_class", 1.0),
        "musicnet_instruments": (ev, fs_hcqt, "instruments", 1.0),
        "musicnet_short": (ev, fs_hcqt, "pitch", 0.25),
        "musicnet_lowrate": (ev, 4.0, "pitch", 1.0),             # 4 frames per second: most notes vanish or last one frame
        "sparse_pitch": (sparse, fs_hcqt, "pitch", 1.0),
        "sparse_short": (sparse, 20.0, "pitch_class", 0.5),
        "dense_error": (dense, fs_hcqt, "pitch", 1.0),          # the reference's own assertion fires: error parity
    }
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for name, (events, fs, kind, shorten) in cases.items():
        n_frames = int(np.floor(events[:, 1].max() * fs)) + 8
        f_hcqt = np.zeros((1, n_frames, 1))
        try:
            out = annot(events.copy(), f_hcqt, fs, annot_type=kind, shorten=shorten)
        except AssertionError as e:
            np.savez_compressed(os.path.join(GOLDEN_DIR, f"annot_{name}.npz"), events=events, fs=np.array(fs),
                                kind=np.array(kind), shorten=np.array(shorten), n_frames=np.array(n_frames),
                                error=np.array(str(e)))
            print(f"{name:22s} events {events.shape[0]:5d} fs {fs:8.4f} frames {n_frames:6d} -> AssertionError: {e}")
            continue
        assert out.dtype == np.float64 and set(np.unique(out)) <= {0.0, 1.0}
        np.savez_compressed(os.path.join(GOLDEN_DIR, f"annot_{name}.npz"), events=events, fs=np.array(fs), kind=np.array(kind),
                            shorten=np.array(shorten), n_frames=np.array(n_frames), shape=np.array(out.shape),
                            bits=np.packbits(out.astype(np.uint8)), hopsize=np.array(hopsize))
        print(f"{name:22s} events {events.shape[0]:5d} fs {fs:8.4f} frames {n_frames:6d} active {int(out.sum()):8d}")


if __name__ == "__main__":
    main()
