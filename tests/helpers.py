"""Shared helpers for the parity tests (test infrastructure; may import oracle/)."""
import glob
import json
import os

import numpy as np
import torch

from multipitch_architectures_amd import nn_models
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.synth import det_fill, synth_batch
from oracle import restate

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))):
        base = os.path.basename(f)[:-4]
        if "__" not in base:            # data_*.npz / metrics_*.npz belong to the section-8(f) tests
            continue
        name, rest = base.split("__")
        B, T = rest.split("_")
        out.append((name.replace("_", ":", 1), int(B[1:]), int(T[1:])))
    return out


def load_golden(name, B, T):
    return np.load(os.path.join(GOLDEN_DIR, f"{name.replace(':', '_')}__B{B}_T{T}.npz"))


def build_model(name, device="cpu"):
    cfg = CONFIGS[name]
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    model.load_state_dict(det_fill(model.state_dict()))
    return model.to(device)


def oracle_kwargs(name):
    cfg = CONFIGS[name]
    return cfg["cls"], dict(cfg["kwargs"])


def sample_idx(n, k=64):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def summarize(t, k=64):
    a = t.detach().double().cpu().numpy().ravel()
    return np.array([a.mean(), a.std(), np.abs(a).max()]), a[sample_idx(a.size, k)].astype(np.float32)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def oracle_forward(name, sd, x, train=False, taps=None, zero_dropout=False):
    cls, kw = oracle_kwargs(name)
    if zero_dropout:
        kw["p_dropout"] = 0.0
    fn = restate.MODELS[cls]
    if zero_dropout and cls == "simple_u_net_doubleselfattn":
        # attention layers are built with a hard-coded 0.2 in the reference; the golden step forces every Dropout.p to 0
        return _saunet_p0(sd, x, train, taps, kw)
    return fn(sd, x, train=train, taps=taps, **kw)


def _saunet_p0(sd, x, train, taps, kw):
    def bott(x5):
        x5 = restate.transformer_enc_layer(x5, sd, "attention1", kw["num_heads"], train, 0.0, kw.get("pos_encoding"))
        return restate.transformer_enc_layer(x5, sd, "attention2", kw["num_heads"], train, 0.0, None)
    return restate._unet(sd, x, train, taps, kw["a_lrelu"], 0.0, kw.get("convdrop", 0), kw.get("residual", False),
                         bottleneck=bott, alt_order=kw.get("alt_order", False))[0]


def oracle_loss(name, res, y):
    if isinstance(res, tuple):
        return restate.punet_loss(res[0], res[1], y)
    return restate.bce_loss(res, y)
