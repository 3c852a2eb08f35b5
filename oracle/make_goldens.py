"""ORACLE tooling -- generates tests/golden/*.npz by importing the reference.

Run ONLY in the build container (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens.py [--only SUBSTR]

It imports ``libdl.nn_models`` from /root/reference, builds each model with the
kwargs of multipitch_architectures_amd/configs.py, loads the deterministic
weight fill (multipitch_architectures_amd/synth.py:det_fill -- so no weight blob
is committed), runs the reference on the seeded synthetic batch and stores
*data only*: outputs, pre-sigmoid logits, per-stage statistics + 64 strided
samples, and for train-mode cases (all dropout p forced to 0, RNG-free) the
loss, gradient norms/samples, BatchNorm running statistics after the step and a
3-step BCELoss+AdamW trajectory.
"""
import argparse
import json
import os
import sys
import unittest.mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
sys.dont_write_bytecode = True

from multipitch_architectures_amd.configs import CONFIGS  # noqa: E402
from multipitch_architectures_amd.synth import det_fill, synth_batch  # noqa: E402

import libdl.nn_models as ref_models  # noqa: E402  (the reference)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# (config, batch, frames, train_step?)
CASES = []
for _name in ("tiny:CNN", "tiny:DRCNN", "tiny:Unet", "tiny:SAUnet", "tiny:SAUnet-res", "tiny:SAUSnet",
              "tiny:BLUnet", "tiny:PUnet"):
    CASES += [(_name, 1, 75, False), (_name, 2, 75, True), (_name, 8, 75, False), (_name, 2, 174, False)]
CASES += [("tiny:SAUnet", 25, 75, True), ("tiny:SAUnet", 50, 75, False), ("tiny:Unet", 3, 100, True)]
CASES += [("tiny:Unet", 32, 75, True)]
CASES += [("tiny:SAUnet-alt", 2, 75, True), ("tiny:SAUnet-alt", 3, 100, False)]     # double_conv(alt_order=True): ELU-BN-Dropout-Conv      # BatchNorm over 32 patches is not chaotic: a tight whole-model gradient check
CASES += [
    ("CNN:XS", 8, 75, True), ("CNN:XS", 8, 174, False),        # BASELINE.json configs[0]
    ("DRCNN:L", 1, 75, False),
    ("Unet:L", 2, 75, True),
    ("SAUnet:L", 2, 75, True), ("SAUnet:L", 2, 174, False), ("SAUnet:L", 25, 75, False),
    ("SAUSnet:L", 2, 75, True),
    ("BLUnet:XXL", 2, 75, True),
    ("PUnet:XL", 2, 75, True), ("PUnet:M", 2, 75, True),
]
# round 3: paper-size train-step goldens for the configurations that only had forward ones (the 70-cout tap-fold
# backward-weight inside DRCNN:L, PUnet:XL's 256/512-channel levels, SAUSnet:L's second attention stage), and one
# non-chaotic (BatchNorm over 32 patches) train case per family for the tight whole-model gradient check
CASES += [("DRCNN:L", 2, 75, True)]
for _name in ("tiny:CNN", "tiny:DRCNN", "tiny:SAUnet", "tiny:SAUSnet", "tiny:BLUnet", "tiny:PUnet"):
    CASES += [(_name, 32, 75, True)]

_real_zeros = torch.zeros


def _zeros_cpu(*a, **kw):
    if "device" in kw and "cuda" in str(kw["device"]):
        kw["device"] = "cpu"      # transformer_enc_layer hard-codes device="cuda:0" (unet_cnns.py:121)
    return _real_zeros(*a, **kw)


def build_reference(cfg_name):
    cfg = CONFIGS[cfg_name]
    with unittest.mock.patch("torch.zeros", _zeros_cpu):
        model = getattr(ref_models, cfg["cls"])(**cfg["kwargs"])
    model.load_state_dict(det_fill(model.state_dict()))
    return model


def sample_idx(n, k=64):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def summarize(t, k=64):
    a = t.detach().double().numpy().ravel()
    return (np.array([a.mean(), a.std(), np.abs(a).max()], dtype=np.float64),
            a[sample_idx(a.size, k)].astype(np.float32))


TAPS = {  # reference module name -> tap name used by oracle/restate.py
    "inc": "x1", "down1": "x2", "down2": "x3", "down3": "x4", "down4": "x5",
    "attention2": "x5b", "attention4": "x4b", "lstm5": "x5b",
    "upconv1": "u1", "upconv2": "u2", "upconv3": "u3", "upconv4": "u4",
    "conv1": "conv1", "conv2": "conv2", "conv3": "conv3", "conv4.3": "logits", "convP": "n_pred",
    "prefilt_list.0": "prefilt0", "prefilt_list.1": "prefilt1", "prefilt_list.2": "prefilt2",
    "prefilt_list.3": "prefilt3",
}


def _to64(model):
    model = model.double()
    for m in model.modules():
        if hasattr(m, "pe") and isinstance(m.pe, torch.Tensor) and not isinstance(m.pe, torch.nn.Parameter):
            m.pe = m.pe.double()
    return model


def _loss(res, y, crit, ce):
    if isinstance(res, tuple):
        n_target = torch.sum(y, dim=-1, keepdims=True).long().squeeze(3)     # exp195f...py:331
        return crit(res[0], y) + ce(res[1], n_target) / 25.0
    return crit(res, y)


def run_case64(cfg_name, B, T, train_step, out):
    """The same case with the reference cast to float64: the rounding-free truth that bounds fp32 noise."""
    model = _to64(build_reference(cfg_name))
    x, y = synth_batch(B, T, seed=1234)
    x, y = x.double(), y.double()
    logits = {}
    h = dict(model.named_modules())["conv4.3"].register_forward_hook(lambda m, i, o: logits.__setitem__("l", o))
    model.eval()
    with torch.no_grad():
        res = model(x)
    out["y64"] = (res[0] if isinstance(res, tuple) else res).numpy()
    out["logits64"] = logits["l"].numpy()
    if isinstance(res, tuple):
        out["n_pred64"] = res[1].numpy()
    h.remove()
    if train_step:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        model.train()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
        losses = []
        for step in range(3):
            res = model(x)
            loss = _loss(res, y, torch.nn.BCELoss(reduction="mean"), torch.nn.CrossEntropyLoss())
            opt.zero_grad()
            loss.backward()
            if step == 0:
                out["train.loss64"] = np.array(loss.item())
                for k, p in model.named_parameters():
                    out[f"grad64.{k}.norm"] = np.array(p.grad.norm().item())
                    out[f"grad64.{k}.absmax"] = np.array(p.grad.abs().max().item())
                    out[f"grad64.{k}.samples"] = p.grad.numpy().ravel()[sample_idx(p.grad.numel(), 16)]
            opt.step()
            losses.append(loss.item())
        out["train.losses64"] = np.array(losses)          # the fp64 trajectory: |losses - losses64| is fp32 chaos
        for k, p in model.named_parameters():
            out[f"p3_64.{k}"] = np.array([p.detach().sum().item(), p.detach().norm().item()])


def run_case(cfg_name, B, T, train_step):
    torch.manual_seed(0)
    model = build_reference(cfg_name)
    x, y = synth_batch(B, T, seed=1234)
    out = {"schema": np.array(json.dumps({k: list(v.shape) for k, v in model.state_dict().items()}))}
    taps = {}
    hooks = []
    mods = dict(model.named_modules())
    for mname, tname in TAPS.items():
        if mname in mods:
            hooks.append(mods[mname].register_forward_hook(
                lambda m, i, o, tname=tname: taps.__setitem__(tname, o)))
    hooks.append(mods["layernorm"].register_forward_hook(
        lambda m, i, o: taps.__setitem__("x_norm", o.transpose(1, 2))))
    model.eval()
    with torch.no_grad():
        res = model(x)
    y_pred = res[0] if isinstance(res, tuple) else res
    out["y"] = y_pred.numpy()
    out["logits"] = taps["logits"].numpy()
    if isinstance(res, tuple):
        out["n_pred"] = res[1].numpy()
    for tname, t in taps.items():
        st, sm = summarize(t)
        out[f"tap.{tname}.stats"] = st
        out[f"tap.{tname}.samples"] = sm
    for h in hooks:
        h.remove()

    if train_step:
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        model.train()
        crit = torch.nn.BCELoss(reduction="mean")
        ce = torch.nn.CrossEntropyLoss()
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                                weight_decay=0.01, amsgrad=False)
        losses = []
        for step in range(3):
            res = model(x)
            if isinstance(res, tuple):
                n_target = torch.sum(y, dim=-1, keepdims=True).long().squeeze(3)
                loss = crit(res[0], y) + ce(res[1], n_target) / 25.0
            else:
                loss = crit(res, y)
            opt.zero_grad()
            loss.backward()
            if step == 0:
                out["train.y"] = (res[0] if isinstance(res, tuple) else res).detach().numpy()
                for k, p in model.named_parameters():
                    g = p.grad
                    out[f"grad.{k}.norm"] = np.array(g.double().norm().item())
                    out[f"grad.{k}.samples"] = g.numpy().ravel()[sample_idx(g.numel(), 16)]
            opt.step()
            if step == 0:
                for k, b in model.named_buffers():
                    if k.endswith(("running_mean", "running_var")):
                        out[f"bn1.{k}"] = b.numpy().copy()
            losses.append(loss.item())
        out["train.losses"] = np.array(losses, dtype=np.float64)
        for k, p in model.named_parameters():
            out[f"p3.{k}"] = np.array([p.detach().double().sum().item(), p.detach().double().norm().item()])
    if B * (T - 74) <= 16 or cfg_name.startswith("tiny") or cfg_name == "CNN:XS":
        run_case64(cfg_name, B, T, train_step, out)
    return out


def case_file(cfg_name, B, T):
    return os.path.join(GOLDEN_DIR, f"{cfg_name.replace(':', '_')}__B{B}_T{T}.npz")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    for cfg_name, B, T, tr in CASES:
        if args.only and args.only not in cfg_name and args.only not in os.path.basename(case_file(cfg_name, B, T)):
            continue
        out = run_case(cfg_name, B, T, tr)
        np.savez_compressed(case_file(cfg_name, B, T), **out)
        print(f"{cfg_name:16s} B={B:<3d} T={T:<4d} train={tr!s:5s} y[min,max]=({out['y'].min():.4f},{out['y'].max():.4f})"
              f" keys={len(out)}", flush=True)


if __name__ == "__main__":
    main()
