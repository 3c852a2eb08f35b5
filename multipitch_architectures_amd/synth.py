"""Synthetic HCQT patches and deterministic weight fills.

Nothing here exists in the reference; it is the build's own generator for
inputs/weights so that goldens never need weight blobs (SURVEY.md §4, §8(c,d)).

* ``synth_batch``   -- X = log(1 + 10*A), A ~ Gamma(0.3, 0.05)  (the reference's log
                      compression, libdl/data_loaders/hcqt_datasets.py:105-106),
                      Y ~ Bernoulli(0.04), layout (B, 6, T, 216) / (B, 1, T-74, 72).
* ``det_fill``      -- per-key PCG64 stream seeded by sha256(key): He-normal for
                      matrices / conv filters, U(0.5, 1.5) for norm gammas and
                      running_var, N(0, 0.1) for biases and running_mean.  With
                      PyTorch's default init every model emits ~0.4953 for every bin
                      (SURVEY.md §4 "trap"), which would make a 1e-4 tolerance vacuous.
"""
import hashlib

import numpy as np
import torch

N_HARM = 6
N_BINS = 216
N_PITCH = 72
CONTEXT = 75


def _rng(tag: str, seed: int = 0) -> np.random.Generator:
    h = hashlib.sha256(f"{seed}:{tag}".encode()).digest()
    return np.random.Generator(np.random.PCG64(int.from_bytes(h[:8], "little")))


def synth_batch(batch: int, frames: int = CONTEXT, seed: int = 1234, n_bins_out: int = N_PITCH):
    """Return (X, Y) float32 CPU tensors: X (B,6,T,216), Y (B,1,T-74,n_bins_out)."""
    g = _rng("hcqt", seed)
    a = g.gamma(shape=0.3, scale=0.05, size=(batch, N_HARM, frames, N_BINS))
    x = np.log1p(10.0 * a).astype(np.float32)
    y = (g.random(size=(batch, 1, frames - (CONTEXT - 1), n_bins_out)) < 0.04).astype(np.float32)
    return torch.from_numpy(x), torch.from_numpy(y)


def _is_gamma(key: str, ndim: int) -> bool:
    if key.endswith("running_var"):
        return True
    if not key.endswith("weight"):
        return False
    leaf_parent = key.rsplit(".", 1)[0].rsplit(".", 1)[-1]
    if "layernorm" in leaf_parent:
        return True
    return ndim == 1      # BatchNorm2d gamma


def det_fill(state_dict, seed: int = 0):
    """Return a new dict with every entry of ``state_dict`` replaced by the deterministic fill."""
    out = {}
    for key, ref in state_dict.items():
        shape = tuple(ref.shape)
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros(shape, dtype=ref.dtype)
            continue
        g = _rng(key, seed)
        if _is_gamma(key, len(shape)):
            v = g.uniform(0.5, 1.5, size=shape)
        elif key.endswith("bias") or key.endswith("running_mean") or "bias_" in key.rsplit(".", 1)[-1] \
                or len(shape) < 2:
            v = g.normal(0.0, 0.1, size=shape)
        else:
            fan_in = int(np.prod(shape[1:]))
            v = g.normal(0.0, np.sqrt(2.0 / fan_in), size=shape)
        out[key] = torch.from_numpy(np.asarray(v, dtype=np.float32)).to(ref.dtype)
    return out


def synth_file(frames=400, n_harm=6, n_bins=216, n_bins_out=72, seed=77):
    """One synthetic "recording": raw (uncompressed) HCQT magnitudes (n_harm, frames, n_bins) float32 ~ Gamma(0.3, 0.05)
    and a binary pitch activity matrix (frames, n_bins_out) float32 ~ Bernoulli(0.04) -- the shapes
    ``dataset_context`` receives after the transpose in exp126a...py:262-263."""
    rng = np.random.Generator(np.random.PCG64(seed))
    inputs = rng.gamma(0.3, 0.05, size=(n_harm, frames, n_bins)).astype(np.float32)
    targets = (rng.random((frames, n_bins_out)) < 0.04).astype(np.float32)
    return inputs, targets


def synth_eval_pair(n_frames=500, n_bins=72, seed=5, quant=None, silent_frames=0, scale=1.0):
    """Targets/predictions for the evaluation measures: Bernoulli(0.04) targets (float32 0/1) and predictions that are
    a noisy logistic function of them (float32 in (0,1)).  ``quant`` rounds predictions to that many levels (ties for
    the ranking measures), ``silent_frames`` leading frames get all-zero targets, ``scale`` < 1 shrinks predictions
    (``scale=0.3`` keeps everything below the 0.4 threshold)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    targ = (rng.random((n_frames, n_bins)) < 0.04).astype(np.float32)
    targ[:silent_frames] = 0
    z = 3.0 * targ - 2.0 + rng.normal(0, 1.2, size=targ.shape)
    pred = (scale / (1.0 + np.exp(-z))).astype(np.float32)
    if quant:
        pred = (np.round(pred * quant) / quant).astype(np.float32)
    return targ, pred
