import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = 256
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
LAYERS = [(16, 128, 75, 216, 15, 1), (16, 128, 75, 216, 15, 0), (16, 16, 75, 216, 15, 0), (32, 16, 75, 216, 15, 0)]
if len(sys.argv) > 1: LAYERS = [LAYERS[int(sys.argv[1])]]
for (ci, co, H, W, k, mode) in LAYERS:
    d = L.ConvDesc(B, ci, H, W, co, k, k, 1, 1, k // 2, k // 2)
    buf = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512)
    w = torch.randn(co, ci, k, k, device="cuda") * 0.02
    n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), mode); wp = torch.empty(n, device="cuda")
    assert lib.mpa_conv2d_pack(ctypes.byref(d), mode, P(w), P(wp), st) == 0
    x = torch.randn(B, ci, H, W, device="cuda"); y = torch.randn(B, co, H, W, device="cuda")
    if mode == 1: f = lambda: lib.mpa_conv2d_bwd_data(ctypes.byref(d), P(y), P(wp), P(x), st)
    else: f = lambda: lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp), None, P(y), 0, ctypes.c_float(0), st)
    assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    fl = 2.0 * B * H * W * ci * co * k * k
    print(f"{ci}->{co} mode {mode}: {ms:7.3f} ms {fl/ms/1e9:6.1f} TF/s  {buf.value.decode()[:100]}")
