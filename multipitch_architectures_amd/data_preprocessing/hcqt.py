"""Pre-processing either side of the HCQT -- drop-in for the functions of
libdl/data_preprocessing/hcqt.py that 01_precompute_features.ipynb / 02_predict_with_pretrained_model.ipynb call.

Built and pinned: ``compute_hopsize_cqt`` (:9-27, scalar arithmetic) and ``compute_annotation_array_nooverlap`` (:205-272, the
note list -> piano roll conversion: a HIP kernel through the C ABI, bit-exact with the reference -- tests/test_gpu_annot.py
against fixtures produced by the reference's own function on the note list it ships).
Built, **parity unpinned**: ``compute_hcqt`` / ``compute_efficient_hcqt`` (:31-164).  In the reference they are thin loops
around ``librosa.cqt`` and ``librosa.estimate_tuning`` (librosa 0.8, third-party, absent from the image: no executable truth
and no fixture to pin a re-implementation to).  Here the published algorithm runs on the GPU -- piptrack / tuning histogram
as kernels, the constant-Q filter bank (librosa's ``filters.constant_q``: Hann windows of length Q sr / f, L1-normalised,
``scale=True``) evaluated *directly at the original sample rate* as a strided GEMM instead of librosa's octave-wise
resampling recursion -- and is checked against ``oracle/restate_hcqt.py``, a float64 restatement of the same definition
(tests/test_gpu_hcqt.py).  Differences from librosa's own output are expected at the level of its resampling /
``sparsity=0.01`` approximations; they have not been measured (they cannot be, here).
``compute_annotation_array`` (:167-202) has no ``return`` in the reference (it yields ``None``) and no caller; it is not
provided.
"""
import ctypes
from fractions import Fraction

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


NOTE_C1_HZ = 32.70319566257483        # librosa.note_to_hz('C1'), the default fmin of the reference's signatures


def _vp(t):
    return ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _as_device_audio(f_audio, device):
    if not torch.cuda.is_available():
        raise RuntimeError("multipitch_architectures_amd: the HCQT kernels need the GPU (no CPU fallback)")
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    y = torch.as_tensor(np.ascontiguousarray(np.asarray(f_audio, dtype=np.float32))) if not torch.is_tensor(f_audio) \
        else f_audio.to(torch.float32)
    y = y.to(device).contiguous()
    if y.dim() != 1 or y.numel() < 2:
        raise ValueError("f_audio must be a mono signal (1-D, at least 2 samples)")
    return y


def _frames_times_basis(ypad, a_off, hop, basis, frames):
    """C[frames][cols] = sum_k ypad[a_off + m hop + k] basis[k][cols]: the strided-A GEMM (overlapping frames, no copy)"""
    K, cols = basis.shape
    C = torch.empty((frames, cols), dtype=torch.float32, device=ypad.device)
    rc = L.load().mpa_gemm(ctypes.c_void_p(ypad.data_ptr() + 4 * a_off), hop, 1, _vp(basis), cols, 1, None, _vp(C), cols, frames,
                           cols, K, 0, 0, _stream())
    L.check(rc, "mpa_gemm")
    return C


def estimate_tuning_device(y, sr=22050, n_fft=2048, bins_per_octave=12, resolution=0.01):
    """librosa.estimate_tuning(y=..., bins_per_octave=...) as restated in oracle/restate_hcqt.py; y: device float32 tensor"""
    lib = L.load()
    n, hop, nb = y.numel(), n_fft // 4, n_fft // 2 + 1
    ypad = torch.empty(n + n_fft, dtype=torch.float32, device=y.device)
    L.check(lib.mpa_reflect_pad(_vp(y), n, n_fft // 2, n_fft // 2, _vp(ypad), _stream()), "mpa_reflect_pad")
    frames = 1 + n // hop
    basis = torch.empty((n_fft, 2 * nb), dtype=torch.float32, device=y.device)
    L.check(lib.mpa_stft_basis(_vp(basis), n_fft, _stream()), "mpa_stft_basis")
    C = _frames_times_basis(ypad, 0, hop, basis, frames)
    S = torch.empty((frames, nb), dtype=torch.float32, device=y.device)
    L.check(lib.mpa_complex_mag(_vp(C), _vp(S), frames * nb, _stream()), "mpa_complex_mag")
    cap = frames * ((nb + 1) // 2)
    pitch = torch.empty(cap, dtype=torch.float64, device=y.device)
    mag = torch.empty(cap, dtype=torch.float32, device=y.device)
    count = torch.zeros(1, dtype=torch.int32, device=y.device)
    L.check(lib.mpa_piptrack(_vp(S), frames, nb, float(sr), n_fft, 150.0, 4000.0, 0.1, _vp(pitch), _vp(mag), _vp(count),
                             _stream()), "mpa_piptrack")
    n_c = int(count.item())
    nbytes = int(lib.mpa_pitch_tuning_workspace(n_c))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=y.device)
    tuning = torch.zeros(1, dtype=torch.float64, device=y.device)
    L.check(lib.mpa_pitch_tuning(_vp(pitch), _vp(mag), n_c, int(bins_per_octave), float(resolution), _vp(tuning), _vp(ws), nbytes,
                                 _stream()), "mpa_pitch_tuning")
    return float(tuning.item())


def _cqt_into(y, sr, hop, fmin, n_bins, bins_per_octave, out, members):
    """| constant-Q transform | of y (direct evaluation) scattered into the HCQT tensor `out` [n_bins_out][frames][n_harm];
    members: (first bin of the slice, harmonic index) per harmonic that takes `out.shape[0]` bins of this transform"""
    lib = L.load()
    n = y.numel()
    frames = 1 + n // hop
    Q = 1.0 / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    r4 = lambda v: (int(v) + 3) // 4 * 4
    P = r4(np.ceil(Q * sr / fmin / 2.0) + 8)
    ypad = torch.empty(n + 2 * P, dtype=torch.float32, device=y.device)
    L.check(lib.mpa_reflect_pad(_vp(y), n, P, P, _vp(ypad), _stream()), "mpa_reflect_pad")
    fac = (ctypes.c_int * len(members))(*[m[0] for m in members])
    hid = (ctypes.c_int * len(members))(*[m[1] for m in members])
    GROUP, COLS = 32, 64
    for g0 in range(0, n_bins, GROUP):
        nb = min(GROUP, n_bins - g0)
        f0 = fmin * 2.0 ** (g0 / bins_per_octave)
        half = int(np.ceil(Q * sr / f0 / 2.0)) + 2
        K0 = r4(half)
        K = (K0 + half + 31) // 32 * 32
        basis = torch.empty((K, COLS), dtype=torch.float32, device=y.device)
        L.check(lib.mpa_cqt_basis(_vp(basis), K, K0, COLS, float(f0), nb, int(bins_per_octave), float(sr), _stream()),
                "mpa_cqt_basis")
        C = _frames_times_basis(ypad, P - K0, hop, basis, frames)
        L.check(lib.mpa_cqt_mag_scatter(_vp(C), frames, COLS, nb, g0, _vp(out), out.shape[0], out.shape[2], fac, hid,
                                        len(members), _stream()), "mpa_cqt_mag_scatter")


def _octave_classes(num_harmonics, num_subharmonics):
    """The (sub)harmonics 1/(S+1) .. 1/2, 1 .. H (plane order of the HCQT tensor) grouped into classes of frequencies
    that differ by whole octaves.  Per class: (lowest member as a float, [(octaves above it, plane index)]); the lowest
    member's CQT, extended upwards by the largest shift, holds every member as a slice bins_per_octave * shift higher.
    What hcqt.py:130-147 arrives at by searching the earlier harmonics for one a power of two below."""
    ratios = [Fraction(1, s + 1) for s in range(num_subharmonics, 0, -1)] + [Fraction(h) for h in range(1, num_harmonics + 1)]
    odd = lambda n: n // (n & -n)
    classes = {}
    for plane, r in enumerate(ratios):
        classes.setdefault((odd(r.numerator), odd(r.denominator)), []).append((r, plane))
    out = []
    for members in classes.values():
        lowest = min(r for r, _ in members)
        out.append((float(lowest), [(int(r / lowest).bit_length() - 1, plane) for r, plane in members]))
    return sorted(out)


def efficient_hcqt_device(f_audio, fs=22050, fmin=NOTE_C1_HZ, fs_hcqt_target=91, bins_per_octave=60, num_octaves=6,
                          num_harmonics=5, num_subharmonics=1, center_bins=True, device=None, tuning=None):
    """compute_efficient_hcqt with the result left on the GPU: (float32 tensor (n_bins, n_frames, harmonics), fs_hcqt, hop)"""
    y = _as_device_audio(f_audio, device)
    num_octaves_eff = num_octaves + np.ceil(np.log2((num_subharmonics + 1)) + np.log2((num_harmonics))).astype(int)   # :112
    hopsize_cqt, fs_cqt = compute_hopsize_cqt(fs_hcqt_target, fs=fs, num_octaves=num_octaves_eff)
    fs_hcqt = fs / hopsize_cqt
    assert np.mod(bins_per_octave, 12) == 0, 'Error: bins_per_octave no multiple of 12'
    bins_per_semitone = int(bins_per_octave / 12)
    if center_bins:
        fmin = fmin / 2 ** ((bins_per_semitone - 1) / (2 * bins_per_octave))
    tuning_est = estimate_tuning_device(y, sr=fs, bins_per_octave=bins_per_octave) if tuning is None else float(tuning)
    fmin_tuned = fmin * 2 ** (tuning_est / bins_per_octave)
    n_frames = np.floor(y.numel() / hopsize_cqt).astype(int) + 1
    n_bins = bins_per_octave * num_octaves
    f_hcqt = torch.zeros((n_bins, int(n_frames), num_harmonics + num_subharmonics), dtype=torch.float32, device=y.device)
    # harmonics whose frequencies are a whole number of octaves apart are slices of one CQT (hcqt.py:130-152)
    for base_h, members in _octave_classes(num_harmonics, num_subharmonics):
        n_bins_curr = (num_octaves + max(shift for shift, _ in members)) * bins_per_octave
        _cqt_into(y, fs, hopsize_cqt, fmin_tuned * base_h, n_bins_curr, bins_per_octave, f_hcqt,
                  [(shift * bins_per_octave, h) for shift, h in members])
    return f_hcqt, fs_hcqt, hopsize_cqt


def compute_efficient_hcqt(f_audio, fs=22050, fmin=NOTE_C1_HZ, fs_hcqt_target=91, bins_per_octave=60, num_octaves=6,
                           num_harmonics=5, num_subharmonics=1, center_bins=True):
    """ Computes an HCQT in an efficient way using the same CQT for multiple-of-two harmonics (hcqt.py:89-164).
    PARITY UNPINNED (module docstring).

    Returns:
        f_hcqt:            HCQT tensor (numpy float64), dimensions "#pitch_bins * #time_frames * #(sub)harmonics"
        fs_hcqt:           resulting HCQT frame rate in Hz
        hopsize_hcqt:      resulting HCQT hopsize in samples
    """
    f_hcqt, fs_hcqt, hop = efficient_hcqt_device(f_audio, fs, fmin, fs_hcqt_target, bins_per_octave, num_octaves,
                                                 num_harmonics, num_subharmonics, center_bins)
    return f_hcqt.cpu().numpy().astype(np.float64), fs_hcqt, hop


def compute_hcqt(f_audio, fs=22050, fmin=NOTE_C1_HZ, fs_hcqt_target=91, bins_per_octave=60, num_octaves=6, num_harmonics=5,
                 num_subharmonics=1, center_bins=True):
    """ Computes a standard HCQT with one individual CQT for each (sub)harmonic (hcqt.py:31-86).  PARITY UNPINNED. """
    y = _as_device_audio(f_audio, None)
    hopsize_cqt, fs_cqt = compute_hopsize_cqt(fs_hcqt_target, fs=fs, num_octaves=num_octaves)
    fs_hcqt = fs / hopsize_cqt
    n_bins = num_octaves * bins_per_octave
    assert np.mod(bins_per_octave, 12) == 0, 'Error: bins_per_octave no multiple of 12'
    bins_per_semitone = int(bins_per_octave / 12)
    if center_bins:
        fmin = fmin / 2 ** ((bins_per_semitone - 1) / (2 * bins_per_octave))
    tuning_est = estimate_tuning_device(y, sr=fs, bins_per_octave=bins_per_octave)
    fmin_tuned = fmin * 2 ** (tuning_est / bins_per_octave)
    n_frames = 1 + y.numel() // hopsize_cqt
    f_hcqt = torch.zeros((n_bins, n_frames, num_harmonics + num_subharmonics), dtype=torch.float32, device=y.device)
    _cqt_into(y, fs, hopsize_cqt, fmin_tuned, n_bins, bins_per_octave, f_hcqt, [(0, num_subharmonics)])
    for n_ha in range(2, num_harmonics + 1):
        _cqt_into(y, fs, hopsize_cqt, n_ha * fmin_tuned, n_bins, bins_per_octave, f_hcqt, [(0, num_subharmonics + n_ha - 1)])
    for n_hs in range(1, num_subharmonics + 1):
        _cqt_into(y, fs, hopsize_cqt, fmin_tuned / (n_hs + 1), n_bins, bins_per_octave, f_hcqt, [(0, num_subharmonics - n_hs)])
    return f_hcqt.cpu().numpy().astype(np.float64), fs_hcqt, hopsize_cqt
