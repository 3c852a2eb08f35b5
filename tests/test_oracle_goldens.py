"""Pins oracle/restate.py (the CPU restatement) against the golden vectors produced by the reference itself
(oracle/make_goldens.py).  CPU only."""
import json

import numpy as np
import pytest
import torch

from helpers import (build_model, golden_cases, load_golden, oracle_forward, oracle_loss, rel_err, sample_idx, summarize)
from multipitch_architectures_amd.synth import synth_batch
from oracle import restate

SLOW = {("SAUnet:L", 25, 75), ("SAUnet:L", 2, 174), ("DRCNN:L", 1, 75), ("PUnet:XL", 2, 75), ("BLUnet:XXL", 2, 75),
        ("SAUSnet:L", 2, 75)}
CASES = golden_cases()


def _ids(c):
    return f"{c[0]}-B{c[1]}-T{c[2]}"


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_schema_matches_reference(case):
    name, B, T = case
    g = load_golden(name, B, T)
    schema = json.loads(str(g["schema"]))
    sd = build_model(name).state_dict()
    assert list(sd.keys()) == list(schema.keys())
    for k, v in sd.items():
        assert list(v.shape) == schema[k], k


@pytest.mark.parametrize("case", CASES, ids=_ids)
def test_oracle_forward_matches_reference(case):
    name, B, T = case
    if case in SLOW:
        pytest.skip("full-size case covered by test_oracle_fullsize (slow marker)")
    _check_forward(name, B, T)


@pytest.mark.slow
@pytest.mark.parametrize("case", [c for c in CASES if c in SLOW], ids=_ids)
def test_oracle_fullsize(case):
    _check_forward(*case)


def _check_forward(name, B, T):
    g = load_golden(name, B, T)
    sd = {k: v.clone() for k, v in build_model(name).state_dict().items()}
    x, _ = synth_batch(B, T)
    taps = {}
    with torch.no_grad():
        res = oracle_forward(name, sd, x, train=False, taps=taps)
    y = res[0] if isinstance(res, tuple) else res
    assert y.shape == g["y"].shape
    # fp32 CPU vs fp32 CPU: different summation orders only
    assert np.abs(y.numpy() - g["y"]).max() < 2e-5
    assert rel_err(taps["logits"].numpy(), g["logits"]) < 1e-4
    if isinstance(res, tuple):
        assert rel_err(res[1].numpy(), g["n_pred"]) < 1e-4
    for key in g.files:
        if key.startswith("tap.") and key.endswith(".samples"):
            tname = key.split(".")[1]
            if tname not in taps:
                continue
            st, sm = summarize(taps[tname])
            ref_st = g[f"tap.{tname}.stats"]
            scale = max(ref_st[2], 1e-6)
            assert np.abs(sm - g[key]).max() / scale < 1e-4, tname
            assert abs(st[1] - ref_st[1]) / max(ref_st[1], 1e-6) < 1e-3, tname


TRAIN_CASES = [c for c in CASES if "train.losses" in load_golden(*c).files]


@pytest.mark.parametrize("case", [c for c in TRAIN_CASES if c not in SLOW and not c[0].endswith(":L")], ids=_ids)
def test_oracle_train_step_matches_reference(case):
    _check_train(*case)


@pytest.mark.slow
@pytest.mark.parametrize("case", [c for c in TRAIN_CASES if c in SLOW or c[0].endswith(":L")], ids=_ids)
def test_oracle_train_step_fullsize(case):
    _check_train(*case)


def _check_train(name, B, T):
    g = load_golden(name, B, T)
    sd, names = restate.split_state(build_model(name).state_dict())
    x, y = synth_batch(B, T)
    state = {}
    losses = []
    for step in range(3):
        res = oracle_forward(name, sd, x, train=True, zero_dropout=True)
        loss = oracle_loss(name, res, y)
        grads = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
        if step == 0:
            yy = res[0] if isinstance(res, tuple) else res
            assert np.abs(yy.detach().numpy() - g["train.y"]).max() < 2e-5
            for k in names:
                gs = g[f"grad.{k}.samples"].astype(np.float64)
                mine = grads[k].numpy().ravel()[sample_idx(grads[k].numel(), 16)].astype(np.float64)
                if f"grad64.{k}.samples" in g.files:
                    r64 = g[f"grad64.{k}.samples"]
                    scale = float(g[f"grad64.{k}.absmax"])
                    tol = 20.0 * np.abs(gs - r64).max() + 5e-3 * scale + 1e-9     # the reference's own fp32 noise
                    assert np.abs(mine - r64).max() <= tol, k
                else:
                    scale = max(np.abs(gs).max(), float(g[f"grad.{k}.norm"]) / np.sqrt(grads[k].numel()))
                    assert np.abs(mine - gs).max() <= 2e-2 * scale + 1e-9, k
        restate.adamw_step({k: sd[k] for k in names}, grads, state, lr=1e-3)
        if step == 0:
            for key in g.files:
                if key.startswith("bn1."):
                    assert rel_err(sd[key[4:]].numpy(), g[key]) < 1e-5, key
        losses.append(float(loss))
    ref_losses = g["train.losses"]
    assert abs(losses[0] - ref_losses[0]) < 2e-5 * max(1.0, abs(ref_losses[0]))
    if "train.losses64" in g.files:     # later steps: only reproducible to the reference's own fp32-vs-fp64 divergence
        chaos = np.abs(ref_losses - g["train.losses64"])
        stable = chaos < 1e-2           # a step where the reference itself diverges by more is not a test of anything
        dev = np.abs(np.array(losses) - g["train.losses64"])
        assert (dev[stable] <= 4.0 * chaos[stable] + 2e-3 * max(1.0, abs(ref_losses[0]))).all(), (losses, list(ref_losses))
        if not stable.all():
            return
    for k in names:
        if k.endswith(("double_conv.0.bias", "double_conv.4.bias")):
            continue
        ref = g[f"p3.{k}"]
        assert abs(float(sd[k].detach().double().norm()) - ref[1]) < 2e-3 * ref[1] + 1.5e-3 * np.sqrt(sd[k].numel()), k


F64_CASES = [c for c in CASES if "y64" in load_golden(*c).files]


# (tiny:SAUnet, B=50) is left to the fp32 tests: its float64 forward alone takes >1 min on 8 cores, and B=25 already pins
# the batch-axis attention in float64 (forward *and* gradients)
F64_SKIP = {("tiny:SAUnet", 50, 75)}
# the batch-32 train goldens added in round 3 (one per family) exist for the tight fp32 gradient floor on the GPU; their
# float64 pass costs 20-40 s each on 8 cores and pins nothing the batch-2 / batch-8 cases of the same families do not pin
# already (the suite has to run in a few minutes; tiny:Unet kept its batch-32 pass until it was timed at 94 s of a 7-9 minute
# suite on a busy host: its batches 1 / 2 / 3 / 8 stay pinned in float64)
F64_SKIP |= {(n, 32, 75) for n in ("tiny:CNN", "tiny:DRCNN", "tiny:SAUnet", "tiny:SAUSnet", "tiny:BLUnet", "tiny:PUnet", "tiny:Unet")}


@pytest.mark.parametrize("case", [c for c in F64_CASES if c not in SLOW and c not in F64_SKIP
                                  and not c[0].endswith((":L", ":XL", ":XXL"))], ids=_ids)
def test_oracle_float64_is_exactly_the_reference(case):
    """In float64 rounding noise vanishes: the restatement must reproduce the reference's forward, loss and every
    parameter gradient to ~1e-10 -- this is what pins the oracle's *algorithm* (batch-axis attention, double
    projections, BN train statistics, bilinear align_corners, LSTM gate order, BCE clamp, CE/25 ...)."""
    name, B, T = case
    g = load_golden(name, B, T)
    base = build_model(name).state_dict()
    sd, names = restate.split_state({k: (v.double() if v.is_floating_point() else v) for k, v in base.items()})
    x, y = synth_batch(B, T)
    x, y = x.double(), y.double()
    taps = {}
    with torch.no_grad():
        res = oracle_forward(name, sd, x, train=False, taps=taps)
    yy = res[0] if isinstance(res, tuple) else res
    assert np.abs(yy.numpy() - g["y64"]).max() < 1e-11
    assert rel_err(taps["logits"].numpy(), g["logits64"]) < 1e-10
    if "train.loss64" in g.files:
        res = oracle_forward(name, sd, x, train=True, zero_dropout=True)
        loss = oracle_loss(name, res, y)
        assert abs(float(loss) - float(g["train.loss64"])) < 1e-12 * max(1.0, abs(float(loss)))
        grads = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
        for k in names:
            mine = grads[k].numpy().ravel()[sample_idx(grads[k].numel(), 16)]
            scale = max(float(g[f"grad64.{k}.absmax"]), 1e-12)
            assert np.abs(mine - g[f"grad64.{k}.samples"]).max() <= 1e-8 * scale + 1e-12, k


def test_blunet_lstm_depth_above_one_raises_like_the_reference():
    """u_net_blstm_varlayers(lstm_depth>1): `lstm4` is built for embed_dim = C*13 but receives the skip x4 with C*27
    features, so the reference's nn.LSTM raises RuntimeError on the first forward (verified by importing the reference:
    "input.size(-1) must be equal to input_size. Expected 416, got 864").  The oracle restates that; the product's
    check is in tests/test_gpu_configs.py."""
    from multipitch_architectures_amd import nn_models
    kw = dict(n_chan_input=6, n_chan_layers=[8, 8, 6, 4], n_bins_in=216, n_bins_out=72, scalefac=16, embed_dim=416,
              hidden_size=208, lstm_depth=2, lstm_number=1)
    model = nn_models.u_net_blstm_varlayers(**kw)                        # constructs, like the reference
    sd = model.state_dict()
    assert "lstm4.blstm.weight_ih_l0" in sd and tuple(sd["lstm4.blstm.weight_ih_l0"].shape) == (4 * 208, 416)
    x, _ = synth_batch(1, 75)
    with pytest.raises(RuntimeError, match="Expected 416, got 864"):
        restate.u_net_blstm_varlayers(sd, x, **kw)
