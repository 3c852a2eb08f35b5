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


# (Cin, Cout, H, W, k, sh, sw, ph, pw, B, slice): SAUnet:L layers at batch 256; DRCNN:L's 70->70 15x15 (98 % of that
# model's FLOPs, couts padded to 80) at its batch 64; the strided head conv2 of BLUnet:XXL (128->200) at batch 256 and of
# PUnet:XL (128->180) at batch 128 -- the bench batches of BASELINE.json's other configurations
LAYERS = [(16, 128, 75, 216, 15, 1, 1, 7, 7, 256, 32), (32, 16, 75, 216, 15, 1, 1, 7, 7, 256, 32),
          (64, 64, 18, 54, 9, 1, 1, 4, 4, 256, 32), (128, 80, 75, 216, 3, 1, 3, 1, 0, 256, 32),
          (128, 128, 4, 13, 3, 1, 1, 1, 1, 256, 32), (70, 70, 75, 216, 15, 1, 1, 7, 7, 64, 16),
          (128, 200, 75, 216, 3, 1, 3, 1, 0, 256, 64), (128, 180, 75, 216, 3, 1, 3, 1, 0, 128, 32)]


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: "x".join(map(str, l)))
def test_conv_full_batch_equals_slices(layer):
    from multipitch_architectures_amd import ops
    Cin, Cout, H, W, k, sh, sw, ph, pw, B, S = layer
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


def test_blstm_layer_full_batch_equals_slices():
    """BLUnet:XXL's bottleneck BiLSTM (input 1664 = 128 channels x 13 bins, hidden 832, T' = 4; unet_cnns.py:232) at
    batch 256: sequences are independent, so the full batch must equal its slices (outputs, input gradients) and the
    slices' parameter gradients must add up -- the GEMM planner picks other tiles / K splits for 256 x 4 rows than for
    32 x 4."""
    from multipitch_architectures_amd import ops
    B, S, T, I, H = 256, 32, 4, 1664, 832
    g = torch.Generator(device="cuda").manual_seed(9)
    k = 1.0 / H ** 0.5
    params = []
    for _ in range(2):
        params += [(torch.rand(4 * H, I, device="cuda", generator=g) * 2 - 1) * k,
                   (torch.rand(4 * H, H, device="cuda", generator=g) * 2 - 1) * k,
                   (torch.rand(4 * H, device="cuda", generator=g) * 2 - 1) * k,
                   (torch.rand(4 * H, device="cuda", generator=g) * 2 - 1) * k]
    params = [p.requires_grad_(True) for p in params]
    x = torch.randn(B, T, I, device="cuda", generator=g)
    xf = x.clone().requires_grad_(True)
    out = ops.blstm_layer(xf, params)
    go = torch.randn(out.shape, device="cuda", generator=g)
    out.backward(go)
    full = [p.grad.clone() for p in params]
    dx_full = xf.grad.clone()
    for p in params:
        p.grad = None
    acc = [torch.zeros_like(f) for f in full]
    for i in range(0, B, S):
        xs = x[i:i + S].clone().requires_grad_(True)
        o = ops.blstm_layer(xs, params)
        assert _rel(o, out[i:i + S]) < 2e-5
        o.backward(go[i:i + S].contiguous())
        assert _rel(xs.grad, dx_full[i:i + S]) < 5e-5
        for a, p in zip(acc, params):
            a += p.grad
            p.grad = None
    for a, f in zip(acc, full):
        assert _rel(a, f) < 5e-5


def test_transformer_layers_at_batch_256_against_the_oracle():
    """SAUnet:L's two transformer_enc_layers on a bottleneck tensor of the bench batch, (256, 128, 4, 13): the attention
    runs over the *batch* axis (Appendix C.1), so batch slices are not a valid check -- the CPU oracle
    (oracle/restate.py::transformer_enc_layer, pinned to the reference) on the very same tensor is.  S = 52 positions,
    E = 128, 8 heads, MLP 8192, sinusoidal PE on the first layer; forward <= 1e-4, input and parameter gradients against
    the oracle's autograd."""
    from oracle import restate
    from multipitch_architectures_amd.nn_models import transformer_enc_layer
    from multipitch_architectures_amd.synth import det_fill
    torch.manual_seed(0)
    l1 = transformer_enc_layer(embed_dim=128, num_heads=8, mlp_dim=8192, p_dropout=0.0, pos_encoding="sinusoidal")
    l2 = transformer_enc_layer(embed_dim=128, num_heads=8, mlp_dim=8192, p_dropout=0.0)
    sd = {}
    for name, layer in (("attention1", l1), ("attention2", l2)):
        layer.load_state_dict(det_fill(layer.state_dict()))
        sd.update({f"{name}.{k}": v.clone().requires_grad_(True) for k, v in layer.state_dict().items()})
    g = torch.Generator().manual_seed(3)
    x = torch.randn(256, 128, 4, 13, generator=g)
    gy = torch.randn(256, 128, 4, 13, generator=g)

    def oracle(dtype):
        sdd = {k: v.detach().to(dtype).requires_grad_(True) for k, v in sd.items()}
        xr = x.detach().clone().to(dtype).requires_grad_(True)
        r = restate.transformer_enc_layer(xr, sdd, "attention1", 8, True, 0.0, "sinusoidal")
        r = restate.transformer_enc_layer(r, sdd, "attention2", 8, True, 0.0, None)
        r.backward(gy.to(dtype))
        return r.detach(), xr.grad, {k: v.grad for k, v in sdd.items()}

    ref, dx_ref, dp_ref = oracle(torch.float32)
    tru, dx_tru, dp_tru = oracle(torch.float64)          # the truth; the fp32 oracle's own error is the yardstick
    l1.cuda().train(); l2.cuda().train()
    xg = x.detach().clone().cuda().requires_grad_(True)
    y = l2(l1(xg))
    y.backward(gy.cuda())
    assert (y.detach().cpu() - ref).abs().max().item() <= 1e-4 * max(1.0, ref.abs().max().item())
    assert _rel(y.detach().cpu().double(), tru) <= max(3 * _rel(ref.double(), tru), 2e-5)
    assert _rel(xg.grad.cpu().double(), dx_tru) <= max(3 * _rel(dx_ref.double(), dx_tru), 1e-4)
    for name, layer in (("attention1", l1), ("attention2", l2)):
        for k, p in layer.named_parameters():
            key = f"{name}.{k}"
            assert _rel(p.grad.cpu().double(), dp_tru[key]) <= max(3 * _rel(dp_ref[key].double(), dp_tru[key]), 2e-4), key
