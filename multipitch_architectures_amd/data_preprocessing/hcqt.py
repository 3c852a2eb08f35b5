"""Pre-processing either side of the HCQT -- drop-in for the functions of
libdl/data_preprocessing/hcqt.py that 01_precompute_features.ipynb / 02_predict_with_pretrained_model.ipynb call.

Built: ``compute_hopsize_cqt`` (:9-27, scalar arithmetic) and ``compute_annotation_array_nooverlap`` (:205-272, the note
list -> piano roll conversion: a HIP kernel through the C ABI, bit-exact with the reference -- tests/test_gpu_annot.py
against fixtures produced by the reference's own function on the note list it ships).
Not built: ``compute_hcqt`` / ``compute_efficient_hcqt`` (:31-164) -- they are thin loops around ``librosa.cqt`` and
``librosa.estimate_tuning`` (third-party, absent from the image: no executable truth to pin a re-implementation to);
they raise ``NotImplementedError``.  ``compute_annotation_array`` (:167-202) has no ``return`` in the reference (it yields
``None``) and no caller; it is not provided.
"""
import ctypes

import numpy as np
import torch

from .. import _lib as L

_KINDS = {"pitch_class": 0, "pitch": 1, "instruments": 2}
_HEIGHT = {"pitch_class": 12, "pitch": 128, "instruments": 1}


def compute_hopsize_cqt(fs_cqt_target, fs=22050, num_octaves=7):
    """ Computes the necessary CQT hopsize to approximate a desired feature rate fs_cqt_target (hcqt.py:9-27).

    Returns: hopsize_cqt (samples), fs_cqt (resulting frame rate in Hz)
    """
    factor = 2 ** (num_octaves - 1)
    hopsize_target = fs / fs_cqt_target
    n = np.round(hopsize_target / factor)
    hopsize_cqt = int(np.max(np.array([1, factor * n])))
    return hopsize_cqt, fs / hopsize_cqt


def annotation_array_nooverlap_device(note_events, n_frames, fs_hcqt, annot_type="pitch_class", shorten=1.0, device=None):
    """The conversion with the result left on the GPU: float64 tensor (rows, n_frames)."""
    if annot_type not in _KINDS:
        raise AssertionError(["annotation type " + str(annot_type) + " not valid!"])
    if not torch.cuda.is_available():
        raise RuntimeError("multipitch_architectures_amd: the annotation kernel needs the GPU (no CPU fallback)")
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    ev = torch.as_tensor(np.ascontiguousarray(np.asarray(note_events, dtype=np.float64))).to(device)
    if ev.dim() != 2 or ev.shape[1] < 3:
        raise ValueError("note_events must be (n_events, >= 3): start_sec, end_sec, pitch[, ...]")
    n = ev.shape[0]
    lib = L.load()
    if n == 0:                               # an empty note list is an all-zero roll (hcqt.py:228, nothing painted)
        return torch.zeros((_HEIGHT[annot_type], int(n_frames)), dtype=torch.float64, device=device)
    out = torch.empty((_HEIGHT[annot_type], int(n_frames)), dtype=torch.float64, device=device)
    nbytes = int(lib.mpa_annotation_workspace(n))
    ws = torch.empty(nbytes // 4, dtype=torch.int32, device=device)
    with torch.cuda.device(device):
        rc = lib.mpa_annotation_array_nooverlap(ctypes.c_void_p(ev.data_ptr()), int(ev.shape[1]), n, float(fs_hcqt),
                                                float(shorten), _KINDS[annot_type], int(n_frames),
                                                ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(ws.data_ptr()), nbytes,
                                                ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.check(rc, "mpa_annotation_array_nooverlap")
    status = int(ws[0].item())              # (synchronises: the reference's assertion is part of the contract)
    if status == 1:
        raise AssertionError("still events of length<1 after correction!")
    if status == 2:
        raise IndexError(f"pitch index out of bounds for axis 0 with size {_HEIGHT[annot_type]}")
    if status == 3:
        raise NotImplementedError("more than 8192 note events vanish at this frame rate: not built")
    return out


def compute_annotation_array_nooverlap(note_events, f_hcqt, fs_hcqt, annot_type='pitch_class', shorten=1.0):
    """ Converts a note event list into a binary np array, assuming a given frame rate (hcqt.py:205-272)

    Args:
        note_events:       np array of note events 'start_sec', 'end_sec', 'pitchclass', 'MIDI_channel'
        f_hcqt:            HCQT tensor, dimensions "#pitch_bins * #time_frames * #(sub)harmonics" (only its length is used)
        fs_hcqt:           resulting HCQT frame rate in Hz
        annot_type:        type of third column: 'pitch' (MIDI pitch), 'pitch_class' (0...11) or 'instruments'
        shorten:           Fraction of duration for shortening note events

    Returns:
        annot_array:       np array (float64) containing binary pitch activity, dimensions "#pitch_bins * #time_frames"
    """
    out = annotation_array_nooverlap_device(note_events, f_hcqt.shape[1], fs_hcqt, annot_type, shorten)
    if shorten != 1.0 and isinstance(note_events, np.ndarray):
        # the reference shortens the caller's array in place (hcqt.py:232-233; its callers pass note_events.copy())
        note_events[:, 1] = note_events[:, 0] + shorten * (note_events[:, 1] - note_events[:, 0])
    return out.cpu().numpy()


def _needs_librosa(name):
    def fn(*args, **kwargs):
        raise NotImplementedError(
            f"{name} wraps librosa.cqt / librosa.estimate_tuning (libdl/data_preprocessing/hcqt.py:31-164); librosa is not "
            "part of this build and a re-implementation would have no executable reference to be pinned to (SURVEY 8 f4)")
    fn.__name__ = name
    return fn


compute_hcqt = _needs_librosa("compute_hcqt")
compute_efficient_hcqt = _needs_librosa("compute_efficient_hcqt")
