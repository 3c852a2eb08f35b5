import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
B = 256
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
d = L.ConvDesc(B, 128, 75, 216, 80, 3, 3, 1, 3, 1, 0)
x = torch.randn(B, 128, 75, 216, device="cuda"); y = torch.randn(B, 80, 75, 72, device="cuda"); w = torch.randn(80, 128, 3, 3, device="cuda") * 0.03
fl = 2.0 * B * 75 * 72 * 128 * 80 * 9
for mode in (0, 1, 2):
    buf = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512)
    if mode < 2:
        n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), mode); wp = torch.empty(n, device="cuda")
        assert lib.mpa_conv2d_pack(ctypes.byref(d), mode, P(w), P(wp), st) == 0
        f = (lambda: lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp), None, P(y), 0, ctypes.c_float(0), st)) if mode == 0 else (lambda: lib.mpa_conv2d_bwd_data(ctypes.byref(d), P(y), P(wp), P(x), st))
    else:
        n = lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d)); ws = torch.empty(n // 4, device="cuda"); dw = torch.empty_like(w); db = torch.empty(80, device="cuda")
        f = lambda: lib.mpa_conv2d_bwd_weight(ctypes.byref(d), P(x), P(y), P(dw), P(db), P(ws), n, st)
    assert f() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    print(f"conv2 mode {mode}: {ms:7.3f} ms {fl/ms/1e9:6.1f} TF/s  {buf.value.decode()[:120]}")
