"""ORACLE (test infrastructure only -- never imported by the product path).  **PARITY UNPINNED.**

CPU restatement (numpy, float64) of the HCQT front-end, `compute_efficient_hcqt` (libdl/data_preprocessing/hcqt.py:89-164).
The reference's arithmetic lives in a third-party dependency that is absent from /root/reference and from this image:
**librosa 0.8.x** (environment.yml:31; `librosa.cqt`, `librosa.estimate_tuning`) on top of resampy's Kaiser resamplers.
No executable truth exists here and the reference ships no fixture for this function, so nothing below is checked against
the reference's numbers; it restates the *published* algorithm:

* `estimate_tuning` (librosa/core/pitch.py): `piptrack` (|STFT|, n_fft 2048, hop 512, Hann window, centred / reflect
  padded; local maxima of the spectrum above 0.1 x the frame's maximum inside [150, 4000) Hz, refined by parabolic
  interpolation) -> pitches whose magnitude is at least the median -> `pitch_tuning`: histogram (100 bins of 0.01) of the
  fractional part of `bins_per_octave * log2(f / 27.5)` folded to [-0.5, 0.5), argmax.
* `cqt` (librosa/core/constantq.py): filters `filters.constant_q` -- Hann-windowed complex exponentials of length
  `Q * sr / f_k`, `Q = 1 / (2^(1/bpo) - 1)`, L1-normalised, response scaled by `length / sqrt(length)` (`scale=True`) --
  applied to the centred, reflect-padded signal at hop `hop_length`.  librosa evaluates this filter bank octave by octave
  on a signal it halves in rate with a Kaiser-windowed resampler (and drops the smallest 1 % of every filter's spectrum:
  `sparsity=0.01`); this restatement and the HIP kernels evaluate the filters **directly at the original rate**, i.e. the
  transform librosa approximates:  C[k, t] = sqrt(N_k) / sum(w_k) * | sum_n y[t hop + n] w_k[n] exp(-2 pi i f_k n / sr) |.
  Differences from librosa's output are expected at the level of its resampling / sparsification error (a few 1e-3
  relative), not at the level of the model's tolerance -- which is why this row stays "parity unpinned".
* the HCQT assembly (which harmonics share a CQT, the slicing) follows hcqt.py:110-164 line by line.
"""
import numpy as np

A440 = 440.0


def hopsize_cqt(fs_cqt_target, fs=22050, num_octaves=7):
    factor = 2 ** (num_octaves - 1)
    n = np.round((fs / fs_cqt_target) / factor)
    hop = int(max(1, factor * n))
    return hop, fs / hop


def reflect_pad(y, n):
    return np.pad(y, n, mode="reflect")


def stft_mag(y, n_fft=2048, hop=512):
    yp = reflect_pad(np.asarray(y, dtype=np.float64), n_fft // 2)
    frames = 1 + (len(yp) - n_fft) // hop
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)              # periodic Hann (scipy get_window, fftbins)
    idx = np.arange(n_fft)[None, :] + hop * np.arange(frames)[:, None]
    return np.abs(np.fft.rfft(yp[idx] * w, axis=1)).T                          # (1 + n_fft/2, frames)


def piptrack(S, sr=22050, n_fft=2048, fmin=150.0, fmax=4000.0, threshold=0.1):
    avg = 0.5 * (S[2:] - S[:-2])
    shift = 2 * S[1:-1] - S[2:] - S[:-2]
    shift = avg / (shift + (np.abs(shift) < np.finfo(shift.dtype).tiny))
    avg = np.pad(avg, ([1, 1], [0, 0]), mode="constant")
    shift = np.pad(shift, ([1, 1], [0, 0]), mode="constant")
    dskew = 0.5 * avg * shift
    freqs = np.arange(1 + n_fft // 2) * sr / n_fft
    freq_mask = ((fmin <= freqs) & (freqs < fmax))[:, None]
    ref = threshold * S.max(axis=0, keepdims=True)
    Sm = S * (S > ref)
    loc = np.zeros_like(S, dtype=bool)                                          # librosa.util.localmax along axis 0
    loc[1:-1] = (Sm[1:-1] > Sm[:-2]) & (Sm[1:-1] >= Sm[2:])
    loc[-1] = Sm[-1] > Sm[-2]
    idx = loc & freq_mask
    pitches = np.where(idx, (np.arange(S.shape[0])[:, None] + shift) * sr / n_fft, 0.0)
    mags = np.where(idx, S + dskew, 0.0)
    return pitches, mags


def pitch_tuning(frequencies, resolution=0.01, bins_per_octave=12):
    f = np.asarray(frequencies)
    f = f[f > 0]
    if f.size == 0:
        return 0.0
    residual = np.mod(bins_per_octave * np.log2(f / (A440 / 16.0)), 1.0)
    residual[residual >= 0.5] -= 1.0
    bins = np.linspace(-0.5, 0.5, int(np.ceil(1.0 / resolution)) + 1)
    counts, edges = np.histogram(residual, bins)
    return float(edges[np.argmax(counts)])


def estimate_tuning(y, sr=22050, n_fft=2048, bins_per_octave=12, resolution=0.01):
    pitch, mag = piptrack(stft_mag(y, n_fft, n_fft // 4), sr, n_fft)
    mask = pitch > 0
    thr = np.median(mag[mask]) if mask.any() else 0.0
    return pitch_tuning(pitch[(mag >= thr) & mask], resolution, bins_per_octave)


def cqt_filter(freq, sr, Q):
    """filters.constant_q for one bin: sample offsets, window, L1 normalisation"""
    ilen = Q * sr / freq
    n = np.arange(-ilen // 2, ilen // 2, dtype=float)
    M = len(n)
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(M) / M)
    return n, w / w.sum(), ilen


def cqt_mag(y, sr, hop, fmin, n_bins, bins_per_octave):
    """| direct constant-Q transform |, (n_bins, 1 + len(y) // hop)"""
    y = np.asarray(y, dtype=np.float64)
    Q = 1.0 / (2.0 ** (1.0 / bins_per_octave) - 1.0)
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bins_per_octave)
    frames = 1 + len(y) // hop
    nmax = int(np.ceil(Q * sr / freqs[0] / 2)) + 2
    yp = reflect_pad(y, nmax) if len(y) > nmax else np.pad(np.pad(y, len(y) - 1, mode="reflect"), nmax, mode="constant")
    out = np.zeros((n_bins, frames))
    for k, f in enumerate(freqs):
        n, g, ilen = cqt_filter(f, sr, Q)
        ker = g * np.exp(-2j * np.pi * f * n / sr)
        idx = (nmax + hop * np.arange(frames))[:, None] + n.astype(np.int64)[None, :]
        out[k] = np.abs(yp[idx] @ ker) * np.sqrt(ilen)
    return out


def efficient_hcqt(y, fs=22050, fmin=32.70319566257483, fs_hcqt_target=91, bins_per_octave=60, num_octaves=6,
                   num_harmonics=5, num_subharmonics=1, center_bins=True, tuning=None):
    """hcqt.py:89-164 with the two librosa calls replaced by the restatements above"""
    eps = np.finfo(float).eps
    num_octaves_eff = num_octaves + int(np.ceil(np.log2(num_subharmonics + 1) + np.log2(num_harmonics)))
    hop, _ = hopsize_cqt(fs_hcqt_target, fs=fs, num_octaves=num_octaves_eff)
    fs_hcqt = fs / hop
    assert bins_per_octave % 12 == 0
    bps = bins_per_octave // 12
    if center_bins:
        fmin = fmin / 2 ** ((bps - 1) / (2 * bins_per_octave))
    if tuning is None:
        tuning = estimate_tuning(y, sr=fs, bins_per_octave=bins_per_octave)
    fmin_tuned = fmin * 2 ** (tuning / bins_per_octave)
    n_frames = int(np.floor(len(y) / hop)) + 1
    n_bins = bins_per_octave * num_octaves
    out = np.zeros((n_bins, n_frames, num_harmonics + num_subharmonics))
    harmonics = [1 / (s + 1) for s in range(num_subharmonics, 0, -1)] + list(range(1, num_harmonics + 1))
    base = np.zeros(len(harmonics))
    done = np.zeros(len(harmonics))
    base[0] = 1 / (num_subharmonics + 1)
    done[0] = 1
    for h in range(1, len(harmonics)):
        nb = 0
        while done[h] < eps:
            b = base[nb]
            if b == 0:
                base[h] = harmonics[h]
                done[h] = 1
            elif np.mod(np.log2(harmonics[h] / b), 1) == 0:
                base[h] = b
                done[h] = 1
            else:
                nb += 1
    for b in np.unique(base):
        members = np.where(base == b)[0]
        add = int(np.ceil(np.log2(harmonics[members.max()] / b)))
        C = cqt_mag(y, fs, hop, fmin_tuned * b, (num_octaves + add) * bins_per_octave, bins_per_octave)
        for h in members:
            fac = int(np.log2(harmonics[h] / b))
            out[:, :, h] = C[fac * bins_per_octave:(fac + num_octaves) * bins_per_octave, :]
    return out, fs_hcqt, hop, tuning
