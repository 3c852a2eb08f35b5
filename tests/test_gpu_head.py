"""The head stage conv2 -- nn.Conv2d(n0, n1, (3,3), stride (1,3), padding (1,0)), unet_cnns.py:538-543,
basic_cnns.py:390-395 -- on its own kernels (csrc/conv_head.hip): forward with the fused LeakyReLU, backward-data and
backward-weight + bias gradient through the same C ABI entry points as every other convolution, against torch's conv2d
in float64 on the CPU.  The channel counts are the ones of the paper's configurations (n0 = 128 / 64 / 70 / 100 / 40,
n1 = 80 / 150 / 180 / 200 / 100 / 70), plus ragged cases: a partial last channel chunk (70 % 8), cout tiles that do not
fill the wave grid, planes smaller than one pixel tile, T = 174."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from multipitch_architectures_amd import _lib as L
from multipitch_architectures_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _head_forward_at_any_batch(setenv_diag):
    """the planner hands small launches of the forward pass to the generic kernel (conv_plan.h: plan_head); the cases here are
    small, so switch that off -- what is under test is the kernel"""
    setenv_diag("MPA_HEAD_FWD_MIN_WGS", "0")


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _plan(case, mode):
    B, Cin, H, W, Cout = case
    d = L.ConvDesc(B, Cin, H, W, Cout, 3, 3, 1, 3, 1, 0)
    buf = ctypes.create_string_buffer(512)
    assert L.load().mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512) == 0
    return buf.value.decode()


# (B, Cin, H, W, Cout) and the kernels expected for (forward, backward-data, backward-weight)
CASES = [
    ((2, 128, 75, 216, 80), ("head_gemm<5,1,4>", "head_gemm<6,4,2>", "head_wgrad<5>")),       # SAUnet:L
    ((1, 128, 75, 216, 150), ("head_gemm<5,2,4>", "head_gemm<6,4,2>", "head_wgrad<5>")),      # Unet:L (10 cout tiles)
    ((1, 128, 75, 216, 180), ("head_gemm<6,2,4>", "head_gemm<6,4,2>", "head_wgrad<4>")),      # PUnet:XL
    ((1, 128, 75, 216, 200), ("head_gemm<7,2,4>", "head_gemm<6,4,2>", "head_wgrad<5>")),      # BLUnet:XXL (13 tiles: 7 + 6)
    ((2, 70, 40, 216, 70), ("head_gemm<5,1,4>", "head_gemm<7,2,4>", "head_wgrad<5>")),        # DRCNN:L channels, 70 % 8 != 0
    ((3, 100, 9, 24, 100), ("head_gemm<7,1,4>", "head_gemm<5,4,2>", "head_wgrad<4>")),        # plane < one pixel tile
    ((1, 40, 174, 216, 100), ("head_gemm<7,1,4>", "head_gemm<4,2,4>", "head_wgrad<4>")),      # T = 174
    ((2, 64, 75, 216, 30), ("fwd<", "head_gemm<6,2,4>", "head_wgrad<2>")),                    # 2 cout tiles: generic forward
    ((5, 128, 5, 12, 64), ("head_gemm<4,1,4>", "head_gemm<6,4,2>", "head_wgrad<4>")),         # smallest width
    ((1, 144, 10, 24, 120), ("head_gemm<4,2,4>", "head_gemm<7,4,2>", "head_wgrad<4>")),       # 8 cout tiles over 2 wave rows; 27 (channel, phase) tiles; 3 channel groups
    ((2, 32, 1, 24, 64), ("head_gemm<4,1,4>", "fwd<", "head_wgrad<4>")),                      # one image row: every halo row is a zero page
]


@pytest.mark.parametrize("case,kernels", CASES, ids=lambda c: "x".join(map(str, c)) if isinstance(c[0], int) else None)
def test_head_conv_matches_float64(dev, case, kernels):
    B, Cin, H, W, Cout = case
    for mode, want in enumerate(kernels):
        assert _plan(case, mode).startswith(want), (mode, _plan(case, mode))
    x = _rand((B, Cin, H, W), 1)
    w = _rand((Cout, Cin, 3, 3), 2, (2.0 / (Cin * 9)) ** 0.5)
    b = _rand((Cout,), 3, 0.1)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    zr = F.conv2d(xr, wr, br, stride=(1, 3), padding=(1, 0))
    yr = F.leaky_relu(zr, 0.3)
    gy = _rand(tuple(yr.shape), 4) * (zr.detach().abs() > 1e-4).float()
    yr.backward(gy.double())
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, (1, 3), (1, 0), ops.ACT_LRELU, 0.3)
    y.backward(gy.to(dev))

    def close(a, ref, rtol, what):
        err = float((a.detach().cpu().double() - ref).abs().max())
        scale = float(ref.abs().max())
        assert err <= rtol * scale, f"{what}: err {err:.3e} vs scale {scale:.3e}"

    close(y, yr.detach(), 2e-5, "y")
    close(xg.grad, xr.grad, 2e-5, "dx")
    close(wg.grad, wr.grad, 5e-5, "dw")
    close(bg.grad, br.grad, 5e-5, "db")


def test_head_conv_is_bit_reproducible_and_has_no_atomics(dev):
    """two runs of forward / backward give identical bits (fixed-order slice reduction, no atomic adds)"""
    case = (2, 128, 75, 216, 80)
    B, Cin, H, W, Cout = case
    x = _rand((B, Cin, H, W), 11).to(dev)
    w = _rand((Cout, Cin, 3, 3), 12, 0.03).to(dev)
    gy = _rand((B, Cout, H, W // 3), 13).to(dev)
    outs = []
    for _ in range(2):
        xg, wg = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = ops.conv2d(xg, wg, None, (1, 3), (1, 0))
        y.backward(gy)
        outs.append((y.detach().clone(), xg.grad.clone(), wg.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_head_conv_switch_off_gives_the_generic_kernels(dev, setenv_diag):
    setenv_diag("MPA_HEAD_OFF", "1")
    assert _plan((2, 128, 75, 216, 80), 0).startswith("fwd<")
    assert _plan((2, 128, 75, 216, 80), 2).startswith("wgrad")


def test_head_forward_goes_to_the_generic_kernel_for_small_launches(dev, setenv_diag):
    setenv_diag("MPA_HEAD_FWD_MIN_WGS", None)
    assert _plan((16, 128, 75, 216, 80), 0).startswith("fwd<")             # 272 workgroups: the generic kernel fills the chip better
    assert _plan((32, 128, 75, 216, 80), 0).startswith("head_gemm<5,1,4>")
    assert _plan((32, 128, 75, 216, 80), 1).startswith("head_gemm<6,4,2>")  # backward passes: at every batch
    assert _plan((32, 128, 75, 216, 80), 2).startswith("head_wgrad<5>")


# ---------------------------------------------------------------------------------------------- tall filters (conv3, T > 75)
# Conv2d(n1, n2, (75,1)) on patches longer than 75 frames (unet_cnns.py:545-549, basic_cnns.py:397-401): forward and
# backward-data on the same GEMM kernel (tap groups of 25 / 15), backward-weight on the generic kernel.
# (B, Cin, H, W, Cout) and the kernels expected for (forward, backward-data)
TALL_CASES = [
    ((2, 80, 174, 72, 50), ("tall_gemm<4,1,4>", "tall_gemm<5,1,4>")),        # SAUnet:L at T = 174: 100 output frames
    ((1, 150, 100, 72, 100), ("tall_gemm<7,1,4>", "tall_gemm<5,2,4>")),      # Unet:L at T = 100: 7 / 10 row tiles, 15-tap groups
    ((1, 200, 90, 72, 150), ("tall_gemm<5,2,4>", "tall_gemm<7,2,4>")),       # BLUnet:XXL channels
    ((2, 32, 76, 24, 64), ("tall_gemm<4,1,4>", "tall_gemm<4,1,4>")),         # two output rows, narrow plane
    ((2, 20, 100, 72, 10), ("fwd<", "fwd<")),                                # CNN:XS: too few channels, the generic kernels
]


@pytest.mark.parametrize("case,kernels", TALL_CASES, ids=lambda c: "x".join(map(str, c)) if isinstance(c[0], int) else None)
def test_tall_conv_matches_float64(dev, case, kernels):
    B, Cin, H, W, Cout = case
    d = L.ConvDesc(B, Cin, H, W, Cout, 75, 1, 1, 1, 0, 0)
    buf = ctypes.create_string_buffer(512)
    for mode, want in enumerate(kernels):
        assert L.load().mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512) == 0
        assert buf.value.decode().startswith(want), (mode, buf.value)
    x = _rand((B, Cin, H, W), 1)
    w = _rand((Cout, Cin, 75, 1), 2, (2.0 / (Cin * 75)) ** 0.5)
    b = _rand((Cout,), 3, 0.1)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    zr = F.conv2d(xr, wr, br)
    yr = F.leaky_relu(zr, 0.3)
    gy = _rand(tuple(yr.shape), 4) * (zr.detach().abs() > 1e-4).float()
    yr.backward(gy.double())
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, (1, 1), (0, 0), ops.ACT_LRELU, 0.3)
    y.backward(gy.to(dev))
    for a, ref, rtol, what in ((y, yr.detach(), 2e-5, "y"), (xg.grad, xr.grad, 2e-5, "dx"), (wg.grad, wr.grad, 5e-5, "dw"),
                               (bg.grad, br.grad, 5e-5, "db")):
        err = float((a.detach().cpu().double() - ref).abs().max())
        assert err <= rtol * float(ref.abs().max()), f"{what}: err {err:.3e} vs scale {float(ref.abs().max()):.3e}"
