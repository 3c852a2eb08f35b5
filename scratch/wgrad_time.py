"""time mpa_conv2d_bwd_weight for the generic-kernel layers of SAUnet:L at batch 256 (and check against torch)"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
layers = [(64, 32, 37, 108, 9), (32, 16, 37, 108, 9), (32, 64, 18, 54, 9), (64, 64, 18, 54, 9), (64, 128, 9, 27, 5),
          (128, 128, 9, 27, 5), (128, 64, 18, 54, 5), (64, 32, 18, 54, 5), (256, 128, 9, 27, 3), (128, 128, 4, 13, 3)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
tot = 0
for (ci, co, H, W, k) in layers:
    d = L.ConvDesc(B, ci, H, W, co, k, k, 1, 1, k // 2, k // 2)
    buf = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), 2, buf, 512)
    x = torch.randn(B, ci, H, W, device="cuda"); dy = torch.randn(B, co, H, W, device="cuda")
    dw = torch.empty(co, ci, k, k, device="cuda"); db = torch.empty(co, device="cuda")
    n = lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d)); ws = torch.empty(n // 4, device="cuda")
    f = lambda: lib.mpa_conv2d_bwd_weight(ctypes.byref(d), P(x), P(dy), P(dw), P(db), P(ws), n, st)
    assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    fl = 2.0 * B * H * W * ci * co * k * k
    # reference on a slice of the batch (full check is slow): dw is linear in the batch, so compare on B0 images
    B0 = min(B, 4)
    d0 = L.ConvDesc(B0, ci, H, W, co, k, k, 1, 1, k // 2, k // 2)
    n0 = lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d0)); ws0 = torch.empty(n0 // 4, device="cuda")
    assert lib.mpa_conv2d_bwd_weight(ctypes.byref(d0), P(x[:B0].contiguous()), P(dy[:B0].contiguous()), P(dw), P(db), P(ws0), n0, st) == 0
    ref = torch.nn.grad.conv2d_weight(x[:B0].double().cpu(), (co, ci, k, k), dy[:B0].double().cpu(), padding=k // 2)
    err = (dw.cpu().double() - ref).abs().max().item() / ref.abs().max().item()
    tot += ms
    print(f"{ci:4d}->{co:4d} {H}x{W} k{k}: {ms:7.3f} ms  {fl/ms/1e9:6.1f} TF/s  relerr {err:.1e}  {buf.value.decode()[:110]}")
print("total", tot)
