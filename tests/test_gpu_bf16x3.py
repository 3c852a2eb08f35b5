"""The opt-in split-bf16 ("bf16x3") convolution path (csrc/conv_bf16x3.hip) through the C ABI, against torch's conv2d in
float64 on the CPU.  Bar: the error of a layer output stays below 1e-4 of the output's largest entry (the products are
hi*hi + hi*lo + lo*hi of bf16 halves with fp32 accumulation: ~1e-5 of the rms expected), i.e. the north-star's 1e-4
forward bound holds with the flag on; gradients likewise.  The exact-fp32 path remains the default."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from multipitch_architectures_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _precision():
    ops.set_conv_precision("bf16x3")
    yield
    ops.set_conv_precision("f32")


def _data(shape, seed, hcqt=False):
    g = torch.Generator().manual_seed(seed)
    if hcqt:      # non-negative, sparse-ish like log-compressed HCQT magnitudes
        return torch.log1p(10 * torch.distributions.Gamma(0.3, 20.0).sample(shape)).float()
    return torch.randn(shape, generator=g)


# (B, Cin, H, W, Cout, kw, pw): the 15-row filters of the models + ragged tiles (rows not a multiple of 13 / 15, widths not
# a multiple of 16, channels not a multiple of 8 / 16), kw < 15
GEOMS = [(2, 16, 75, 216, 128, 15, 7), (2, 32, 37, 108, 32, 15, 7), (3, 6, 75, 216, 16, 15, 7), (1, 20, 30, 50, 20, 15, 7),
         (2, 8, 16, 16, 8, 15, 7), (1, 70, 26, 40, 70, 15, 7), (2, 16, 40, 100, 16, 13, 6), (1, 3, 15, 17, 5, 15, 0)]


# 9-row filters (forward and backward-data on the bf16x3 path; their backward-weight stays exact fp32):
# (B, Cin, H, W, Cout) of the models' 9x9 layers + ragged ones
GEOMS9 = [(2, 32, 18, 54, 64), (2, 64, 37, 108, 32), (3, 32, 37, 108, 16), (1, 20, 25, 40, 12), (2, 8, 9, 16, 8)]


@pytest.mark.parametrize("geom", GEOMS9, ids=lambda g: "x".join(map(str, g)))
def test_nine_row_filters(dev, geom):
    B, Cin, H, W, Cout = geom
    x = _data((B, Cin, H, W), 21, hcqt=True)
    w = _data((Cout, Cin, 9, 9), 22) / np.sqrt(Cin * 81)
    b = _data((Cout,), 23)
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, b.to(dev), (1, 1), (4, 4))
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, b.double(), padding=4)
    assert (y.detach().cpu().double() - ref.detach()).abs().max().item() <= 1e-4 * max(ref.abs().max().item(), 1.0)
    dy = _data(tuple(ref.shape), 24)
    y.backward(dy.to(dev))
    ref.backward(dy.double())
    assert (xd.grad.cpu().double() - xr.grad).abs().max().item() <= 1e-4 * max(xr.grad.abs().max().item(), 1.0)
    assert (wd.grad.cpu().double() - wr.grad).abs().max().item() <= 1e-4 * max(wr.grad.abs().max().item(), 1.0)


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_conv_forward_backward_data_match_float64(dev, geom):
    B, Cin, H, W, Cout, kw, pw = geom
    x = _data((B, Cin, H, W), 1, hcqt=True)
    w = _data((Cout, Cin, 15, kw), 2) / np.sqrt(Cin * 15 * kw)
    b = _data((Cout,), 3)
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, b.to(dev), (1, 1), (7 if pw else 0, pw))
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=(7 if pw else 0, pw))
    assert y.shape == ref.shape
    err = (y.detach().cpu().double() - ref).abs().max().item()
    assert err <= 1e-4 * max(ref.abs().max().item(), 1.0), f"forward error {err:.3e} (max |y| {ref.abs().max():.3f})"
    dy = _data(tuple(ref.shape), 4)
    y.backward(dy.to(dev))
    xr = x.double().requires_grad_(True)
    F.conv2d(xr, w.double(), None, padding=(7 if pw else 0, pw)).backward(dy.double())
    gerr = (xd.grad.cpu().double() - xr.grad).abs().max().item()
    assert gerr <= 1e-4 * max(xr.grad.abs().max().item(), 1.0), f"backward-data error {gerr:.3e}"


@pytest.mark.parametrize("geom", GEOMS, ids=lambda g: "x".join(map(str, g)))
def test_conv_backward_weight_and_bias_match_float64(dev, geom):
    B, Cin, H, W, Cout, kw, pw = geom
    x = _data((B, Cin, H, W), 11, hcqt=True)
    w = _data((Cout, Cin, 15, kw), 12) / np.sqrt(Cin * 15 * kw)
    b = _data((Cout,), 13)
    wd, bd = w.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    y = ops.conv2d(x.to(dev), wd, bd, (1, 1), (7 if pw else 0, pw))
    dy = _data(tuple(y.shape), 14)
    y.backward(dy.to(dev))
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    F.conv2d(x.double(), wr, br, padding=(7 if pw else 0, pw)).backward(dy.double())
    werr = (wd.grad.cpu().double() - wr.grad).abs().max().item()
    assert werr <= 1e-4 * max(wr.grad.abs().max().item(), 1.0), f"backward-weight error {werr:.3e} (max {wr.grad.abs().max():.3f})"
    berr = (bd.grad.cpu().double() - br.grad).abs().max().item()
    assert berr <= 1e-4 * max(br.grad.abs().max().item(), 1.0), f"bias gradient error {berr:.3e}"


def test_fused_activation_and_batchnorm_partials(dev):
    B, Cin, H, W, Cout = 2, 16, 37, 108, 32
    x, w, b = _data((B, Cin, H, W), 5, hcqt=True), _data((Cout, Cin, 15, 15), 6) / 60.0, _data((Cout,), 7)
    y = ops.conv2d(x.to(dev), w.to(dev), b.to(dev), (1, 1), (7, 7), ops.ACT_LRELU, 0.3).cpu()
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=7), 0.3)
    assert (y.double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    y2, partials = ops.conv2d_stats(x.to(dev), w.to(dev), b.to(dev), (1, 1), (7, 7))
    ref2 = F.conv2d(x.double(), w.double(), b.double(), padding=7)
    s = partials.cpu().double().sum(0)
    np.testing.assert_allclose(s[:, 0].numpy(), ref2.sum((0, 2, 3)).numpy(), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(s[:, 1].numpy(), (ref2 ** 2).sum((0, 2, 3)).numpy(), rtol=1e-4)


def test_layers_without_a_bf16x3_kernel_keep_the_exact_path(dev):
    x, w = _data((2, 8, 20, 24), 8), _data((8, 8, 3, 3), 9)
    y = ops.conv2d(x.to(dev), w.to(dev), None, (1, 1), (1, 1)).cpu()
    ops.set_conv_precision("f32")
    y0 = ops.conv2d(x.to(dev), w.to(dev), None, (1, 1), (1, 1)).cpu()
    assert torch.equal(y, y0)
