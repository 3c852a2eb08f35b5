"""oracle/restate_hcqt.py (PARITY UNPINNED: no librosa, no reference fixture) checked against what *can* be checked on the CPU:
analytic properties of the published algorithm it restates, and the reference's own scalar helper."""
import numpy as np

from oracle import restate_hcqt as R

SR = 22050


def test_a_sinusoid_peaks_at_its_bin_with_magnitude_sqrt_n_a_over_2():
    f, A = 440.0, 0.8
    y = A * np.sin(2 * np.pi * f * np.arange(SR) / SR)
    C = R.cqt_mag(y, SR, 512, 110.0, 108, 36)
    mid = C.shape[1] // 2
    assert C[:, mid].argmax() == 72                                       # 110 Hz * 2^(72/36) = 440 Hz
    Q = 1 / (2 ** (1 / 36) - 1)
    np.testing.assert_allclose(C[72, mid], np.sqrt(Q * SR / f) * A / 2, rtol=2e-3)
    assert C.shape == (108, 1 + len(y) // 512)


def test_tuning_estimate_recovers_a_detuned_tone():
    for cents_of_bin in (0.17, -0.28, 0.0):                               # fraction of a 1/36-octave bin
        f = 440.0 * 2 ** (cents_of_bin / 36)
        y = np.sin(2 * np.pi * f * np.arange(SR) / SR)
        # (parabolic interpolation of a Hann peak is biased by a few hundredths of a 1/36-octave bin -- librosa's too)
        assert abs(R.estimate_tuning(y, sr=SR, bins_per_octave=36) - cents_of_bin) <= 0.07


def test_hcqt_assembly_shares_cqts_between_octave_related_harmonics():
    y = np.sin(2 * np.pi * 330.0 * np.arange(8192) / SR)
    H, fs_h, hop, _ = R.efficient_hcqt(y, fs=SR, fmin=110.0, fs_hcqt_target=50, bins_per_octave=12, num_octaves=2, num_harmonics=4,
                                       num_subharmonics=1, center_bins=False, tuning=0.0)
    assert H.shape == (24, 1 + 8192 // hop, 5) and hop == R.hopsize_cqt(50, fs=SR, num_octaves=5)[0]
    mid = H.shape[1] // 2
    # harmonic h's CQT starts at h * fmin: the 330 Hz tone sits log2(330 / (h fmin)) octaves up
    for h_idx, h in enumerate((0.5, 1, 2, 3, 4)):
        expect = 12 * np.log2(330.0 / (110.0 * h))
        if 0 <= expect < 24:
            assert abs(int(H[:, mid, h_idx].argmax()) - round(expect)) <= 1, (h, expect)
