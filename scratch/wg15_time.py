import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = 256
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (ci, co, H, W, k) in [(16, 128, 75, 216, 15), (32, 16, 75, 216, 15), (16, 16, 75, 216, 15), (32, 32, 37, 108, 15)]:
    d = L.ConvDesc(B, ci, H, W, co, k, k, 1, 1, 7, 7)
    buf = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), 2, buf, 512)
    x = torch.randn(B, ci, H, W, device="cuda"); dy = torch.randn(B, co, H, W, device="cuda")
    dw = torch.empty(co, ci, k, k, device="cuda"); db = torch.empty(co, device="cuda")
    n = lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d)); ws = torch.empty(n // 4, device="cuda")
    f = lambda: lib.mpa_conv2d_bwd_weight(ctypes.byref(d), P(x), P(dy), P(dw), P(db), P(ws), n, st)
    assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"{ci}->{co} {H}x{W}: {ms:7.3f} ms {2.0*B*H*W*ci*co*k*k/ms/1e9:6.1f} TF/s  {buf.value.decode()[:90]}")
