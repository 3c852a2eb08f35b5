"""ORACLE -- test infrastructure only.  Never imported by the product path.

CPU restatement of the reference's patch extraction + augmentation
(``libdl/data_loaders/hcqt_datasets.py:67-141`` ``dataset_context.__getitem__`` and the
time-scaling-free part of ``dataset_context_segm.__getitem__`` ``:199-289``).

Structure differs from the reference on purpose: the random decisions are *drawn first* into a ``draws`` record
(``draw_augmentation``; same torch-CPU generator calls in the same order as the reference, so that with the same
``torch.manual_seed`` the record equals what the reference would have drawn) and the arithmetic is a pure function of
(file tensors, index, draws) (``apply_patch``).  The GPU kernel is fed the same ``draws`` and must reproduce
``apply_patch``; the goldens in tests/golden/data_*.npz (made by oracle/make_goldens_data.py from the imported
reference) pin ``draw_augmentation`` + ``apply_patch`` together.

Reference quirks restated, not fixed:
* the +-0.5-bin tuning average is computed from the pre-update row (RHS evaluated before the in-place write, :114-118);
* edge bins exposed by a roll are refilled with |N(0,1e-4)| *after* the roll (:121-124, :131-135);
* the 12-bin (pitch-class) target is rolled circularly without zero-fill (:136-137);
* the random-EQ parabola is re-drawn until it is non-negative everywhere (:83-97).
Not restated: when ``inputs`` is already float32 and random-EQ is off, the reference's ``X += noise`` writes into the
file tensor itself (``.type(FloatTensor)`` is then a view, :75,101); no experiment uses that combination.
"""
import math

import torch

N_BINS = 216  # hard-coded in the reference's EQ (:87,94)


def harmonic_offset(h):
    """:90-93 -- harmonic index -> bin offset of its EQ centre (3 bins per semitone)."""
    return -36 if h == 0 else int(36 * math.log2(h))


def eq_curve(alpha, beta, n_harm):
    """(n_harm, 216) float32 gain, 1 - 2e-6*alpha*(f - (beta - offset_h))^2  (:94)."""
    f = torch.arange(N_BINS)
    rows = []
    for h in range(n_harm):
        centre = beta - harmonic_offset(h)
        rows.append(1 - (2e-6 * torch.tensor([alpha]) * (f - centre) ** 2))
    return torch.stack(rows).to(torch.float32)


def n_frames(params):
    return 2 * (params["context"] // 2) + params.get("seglength", 1)


def draw_augmentation(params, n_harm, frames):
    """Draw every random quantity of one __getitem__ call in the reference's order."""
    d = {"alpha": 0, "beta": 0, "tune2": 0, "transp": 0, "n1": None, "n2": None, "n3": None}
    shape = (n_harm, frames, N_BINS)
    if params.get("aug:randomeq"):
        while True:
            alpha = int(torch.randint(1, params["aug:randomeq"] + 1, (1,)))
            beta = int(torch.randint(0, N_BINS, (1,)))
            if float(eq_curve(alpha, beta, n_harm).min()) >= 0:
                break
        d["alpha"], d["beta"] = alpha, beta
    if params.get("aug:noisestd"):
        d["n1"] = torch.normal(mean=torch.zeros(shape), std=params["aug:noisestd"] * torch.ones(shape))
    if params.get("aug:tuning"):
        d["tune2"] = int(torch.randint(-2, 3, (1,)))            # in half bins
        if d["tune2"] != 0:
            e = (n_harm, frames, 1)
            d["n2"] = torch.normal(mean=torch.zeros(e), std=1e-4 * torch.ones(e))
    if params.get("aug:transpsemitones"):
        t = params["aug:transpsemitones"]
        d["transp"] = int(torch.randint(-t, t + 1, (1,)))
        if d["transp"] != 0:
            e = (n_harm, frames, 3 * abs(d["transp"]))
            d["n3"] = torch.normal(mean=torch.zeros(e), std=1e-4 * torch.ones(e))
    return d


def apply_patch(inputs, targets, params, index, d):
    """Deterministic part: window, EQ, noise+abs, log compression, tuning shift, transposition."""
    half = params["context"] // 2
    seg = params.get("seglength", 1)
    start = index * params["stride"]
    X = inputs[:, start:start + 2 * half + seg, :].to(torch.float32).clone()
    if "seglength" in params:
        y = targets[start + half:start + half + seg, :].to(torch.float32)[None, None].clone()
    else:
        y = targets[start + half, :].to(torch.float32)[None, None].clone()
    if params.get("aug:randomeq"):
        X = eq_curve(d["alpha"], d["beta"], X.shape[0])[:, None, :] * X
    if params.get("aug:noisestd"):
        X = torch.abs(X + d["n1"])
    if params.get("compression") is not None:
        X = torch.log(1 + params["compression"] * X)
    if params.get("aug:tuning") and d["tune2"] != 0:
        s2 = d["tune2"]
        old = X
        X = old.clone()
        if s2 == 1:
            X[:, :, 1:] = (old[:, :, :-1] + old[:, :, 1:]) / 2
        elif s2 == -1:
            X[:, :, :-1] = (old[:, :, :-1] + old[:, :, 1:]) / 2
        else:
            X = torch.roll(old, s2 // 2, -1)
        if s2 > 0:
            X[:, :, :1] = torch.abs(d["n2"])
        else:
            X[:, :, -1:] = torch.abs(d["n2"])
    if params.get("aug:transpsemitones"):
        p = d["transp"]
        X = torch.roll(X, 3 * p, -1)
        yr = torch.roll(y, p, -1)
        if p > 0:
            X[:, :, :3 * p] = torch.abs(d["n3"])
            yr[:, :, :p] = 0
        elif p < 0:
            X[:, :, 3 * p:] = torch.abs(d["n3"])
            yr[:, :, p:] = 0
        if y.shape[-1] == 12:
            yr = torch.roll(y, p, -1)
        y = yr
    return X, y


def dataset_len(inputs, params):
    """:62-64 / :195-197."""
    if "seglength" in params:
        return (inputs.shape[1] - params["context"] - params["seglength"] + params["stride"]) // params["stride"]
    return (inputs.shape[1] - params["context"]) // params["stride"]


def context_patch(inputs, targets, params, index):
    """One reference-equivalent __getitem__ (consumes the global torch CPU generator)."""
    d = draw_augmentation(params, inputs.shape[0], n_frames(params))
    X, y = apply_patch(inputs, targets, params, index, d)
    return X, y, d
