"""The 8 U-Net variants (and basic_cnn_pool) the reference exports but no experiment script uses (re-compositions of the hot-path blocks,
nn_models/unet_cnns.py bottom): HIP classes against tests/golden/xcls-*.npz, produced by the reference classes themselves
(oracle/make_goldens_variants.py) -- evaluation outputs <= 1e-4, one train-mode step: loss and every parameter gradient."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from multipitch_architectures_amd import nn_models
from multipitch_architectures_amd.configs import VARIANT_CONFIGS
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.nn_models.layers import Dropout
from multipitch_architectures_amd.synth import det_fill, synth_batch

pytestmark = pytest.mark.gpu
GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = sorted(VARIANT_CONFIGS)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _sample_idx(n, k=16):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


@pytest.mark.parametrize("name", NAMES)
def test_variant_matches_the_reference(dev, name):
    g = np.load(os.path.join(GOLDEN_DIR, f"xcls-{name}.npz"))
    model = getattr(nn_models, name)(**VARIANT_CONFIGS[name])
    assert {k: list(v.shape) for k, v in model.state_dict().items()} == json.loads(str(g["schema"]))
    model.load_state_dict(det_fill(model.state_dict()))
    model.to(dev).eval()
    B, T = int(g["B"]), int(g["T"])
    x, y = synth_batch(B, T, seed=1234)
    x, y = x.to(dev), y.to(dev)
    with torch.no_grad():
        res = model(x)
    two = isinstance(res, tuple)
    yy = (res[0] if two else res).cpu().numpy()
    lsm = "logsoftmax" in name           # log-probabilities (B, n_ch_out, 1, bins): no BCE, argmax over the channel axis
    assert yy.shape == g["y"].shape
    assert np.abs(yy - g["y"]).max() <= 1e-4
    if lsm:
        assert (yy.argmax(1) == g["y"].argmax(1)).all() and np.abs(np.exp(yy).sum(1) - 1.0).max() <= 1e-5
    else:
        assert (yy.reshape(B, -1, 72).argmax(-1) == g["y"].reshape(B, -1, 72).argmax(-1)).all()
    if two:
        ref = g["n_pred"]
        assert np.abs(res[1].cpu().numpy() - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())
    for m in model.modules():
        if isinstance(m, Dropout):
            m.p = 0.0
    model.train()
    res = model(x)
    if not two and res.dim() == 5:       # simple_u_net_doubleselfattn_transenc: (B,1,1,T-74,72)
        res = res.squeeze(1)
    if lsm:      # the linear loss of oracle/make_goldens_variants.py: variant_loss (test-side torch arithmetic on the HIP output)
        n = res.numel()
        wpat = torch.cos(torch.arange(n, dtype=torch.float64) * 0.37).reshape(res.shape).float().to(dev)
        loss = (res * wpat).mean()
    else:
        loss = BCELoss()(res[0] if two else res, y)
        if two:
            loss = loss + res[1].mean() / 25.0
    loss.backward()
    assert abs(float(loss) - float(g["train.loss"])) <= 2e-5 * max(1.0, float(g["train.loss"]))
    # judged against the reference run in float64 with the reference's own fp32 error as the yardstick (train-mode BatchNorm
    # at batch 3 amplifies rounding, tests/test_gpu_models.py): 20 x that noise + 2 % of the gradient's largest entry
    for k, p in model.named_parameters():
        if k.endswith(("double_conv.0.bias", "double_conv.4.bias")):
            continue        # conv bias in front of a BatchNorm: the true gradient is exactly 0, both sides hold rounding noise
        if f"grad.{k}.samples" not in g.files:
            assert p.grad is None, k          # constructed but unused upstream as well (attention_time3..6)
            continue
        r32 = g[f"grad.{k}.samples"].astype(np.float64)
        r64 = g[f"grad64.{k}.samples"]
        mine = p.grad.detach().cpu().numpy().ravel()[_sample_idx(p.numel())].astype(np.float64)
        scale = float(g[f"grad64.{k}.absmax"])
        tol = 20.0 * np.abs(r32 - r64).max() + 2e-2 * scale + 1e-9
        assert np.abs(mine - r64).max() <= tol, (k, np.abs(mine - r64).max(), tol)


def test_transformer_temporal_enc_layer_matches_the_reference(dev):
    """the exported layer on its own (B, C, T', F') interface (unet_cnns.py:162-217) against the reference layer run in
    float64 (oracle/make_goldens_variants.py: layer_goldens): output, input gradient and every parameter gradient"""
    g = np.load(os.path.join(GOLDEN_DIR, "xlayer-transformer_temporal_enc_layer.npz"))
    kwargs = json.loads(str(g["kwargs"]))
    layer = nn_models.transformer_temporal_enc_layer(**kwargs)
    assert {k: list(v.shape) for k, v in layer.state_dict().items()} == json.loads(str(g["schema"]))
    layer.load_state_dict(det_fill(layer.state_dict()))
    layer.to(dev).train()
    x = torch.from_numpy(g["x"]).float().to(dev).requires_grad_(True)
    y = layer(x)
    n = y.numel()
    w = torch.cos(torch.arange(n, dtype=torch.float64) * 0.37).reshape(y.shape).float().to(dev)
    (y * w).sum().backward()

    def close(a, ref, what):
        a, ref = a.detach().cpu().double().numpy(), np.asarray(ref, dtype=np.float64)
        assert a.shape == ref.shape, what
        assert np.abs(a - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-6), (what, np.abs(a - ref).max(), np.abs(ref).max())
    close(y, g["y"], "y")
    close(x.grad, g["dx"], "dx")
    for k, p in layer.named_parameters():
        close(p.grad, g[f"grad.{k}"], k)
