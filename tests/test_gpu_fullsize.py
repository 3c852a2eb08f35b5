"""GPU: the kernels at BASELINE.json's full sizes (batch 256 / 128), checked through size-independent properties:
a convolution over the whole batch must equal the same convolution over batch slices (forward, backward-data) and the
sum of the slices' weight gradients (backward-weight).  Whole batch and slices get *different plans* (tile shapes,
cout tiling, K splits, XCD order, channel split), so this cross-validates the planner's full-size choices, which the
small-batch parity tests never exercise."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize("layer", [(16, 128, 75, 216, 15, 1, 1, 7, 7), (32, 16, 75, 216, 15, 1, 1, 7, 7),
                                   (64, 64, 18, 54, 9, 1, 1, 4, 4), (128, 80, 75, 216, 3, 1, 3, 1, 0),
                                   (128, 128, 4, 13, 3, 1, 1, 1, 1)],
                         ids=lambda l: "x".join(map(str, l)))
def test_conv_full_batch_equals_slices(layer):
    from multipitch_architectures_amd import ops
    Cin, Cout, H, W, k, sh, sw, ph, pw = layer
    B, S = 256, 32
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, Cin, H, W, device="cuda", generator=g)
    w = (torch.randn(Cout, Cin, k, k, device="cuda", generator=g) / (Cin * k * k) ** 0.5).requires_grad_(True)
    b = (torch.randn(Cout, device="cuda", generator=g) * 0.1).requires_grad_(True)
    xf = x.clone().requires_grad_(True)
    y = ops.conv2d(xf, w, b, (sh, sw), (ph, pw), ops.ACT_NONE, 0.0)
    gy = torch.randn(y.shape, device="cuda", generator=g)
    y.backward(gy)
    dw_full, db_full, dx_full = w.grad.clone(), b.grad.clone(), xf.grad.clone()
    w.grad = None; b.grad = None
    dw_sum, db_sum = torch.zeros_like(dw_full), torch.zeros_like(db_full)
    for i in range(0, B, S):
        xs = x[i:i + S].clone().requires_grad_(True)
        ys = ops.conv2d(xs, w, b, (sh, sw), (ph, pw), ops.ACT_NONE, 0.0)
        assert _rel(ys, y[i:i + S]) < 1e-5          # different channel-chunk sizes: different summation order
        ys.backward(gy[i:i + S].contiguous())
        assert _rel(xs.grad, dx_full[i:i + S]) < 2e-5
        dw_sum += w.grad; db_sum += b.grad
        w.grad = None; b.grad = None
    assert _rel(dw_sum, dw_full) < 2e-5 and _rel(db_sum, db_full) < 2e-5


def test_unet_full_batch_equals_slices():
    """Unet:L (no batch-axis attention) in eval mode at BASELINE's batch 128 against four slices of 32."""
    from helpers import build_model
    from multipitch_architectures_amd.synth import synth_batch
    dev = torch.device("cuda:0")
    model = build_model("Unet:L", dev).eval()
    x, _ = synth_batch(128, 75)
    x = x.to(dev)
    with torch.no_grad():
        full = model(x)
        for i in range(0, 128, 32):
            part = model(x[i:i + 32].contiguous())
            assert (part - full[i:i + 32]).abs().max().item() < 1e-5
            assert torch.equal(part.argmax(-1), full[i:i + 32].argmax(-1))
