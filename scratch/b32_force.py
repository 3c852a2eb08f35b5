"""time forward-kernel tile variants (MPA_FWD_FORCE is read per call) for the layers that lose most at batch 32"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (ci, co, H, W, k, mode) in [(32, 16, 75, 216, 15, 0), (16, 128, 75, 216, 15, 1), (6, 16, 75, 216, 15, 0), (16, 16, 75, 216, 15, 0)]:
    d = L.ConvDesc(B, ci, H, W, co, k, k, 1, 1, k // 2, k // 2)
    x = torch.randn(B, ci, H, W, device="cuda"); y = torch.randn(B, co, H, W, device="cuda"); w = torch.randn(co, ci, k, k, device="cuda") * 0.03
    buf = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512)
    n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), mode); wp = torch.empty(n, device="cuda")
    assert lib.mpa_conv2d_pack(ctypes.byref(d), mode, P(w), P(wp), st) == 0
    f = (lambda: lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp), None, P(y), 0, ctypes.c_float(0), st)) if mode == 0 else (lambda: lib.mpa_conv2d_bwd_data(ctypes.byref(d), P(y), P(wp), P(x), st))
    assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{ci}->{co} mode {mode}: {ms:7.3f} ms {2.0*B*H*W*ci*co*k*k/ms/1e9:6.1f} TF/s  {buf.value.decode()[:110]}")
