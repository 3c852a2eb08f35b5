"""70 -> 70 15x15 (DRCNN:L) forward / backward-data with and without the cout remainder fold: python scratch/fold_time.py [B]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Cin = Cout = int(sys.argv[2]) if len(sys.argv) > 2 else 70
H, W = 75, 216
lib = L.load(); dev = torch.device("cuda:0")
d = L.ConvDesc(B, Cin, H, W, Cout, 15, 15, 1, 1, 7, 7)
P = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 15, 15, device=dev) * 0.01
y = torch.empty(B, Cout, H, W, device=dev); gy = torch.randn_like(y); dx = torch.empty_like(x)
fl = 2.0 * B * H * W * Cout * Cin * 225
def bank(mode):
    n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), mode); assert n > 0, n
    t = torch.empty(n, device=dev); assert lib.mpa_conv2d_pack(ctypes.byref(d), mode, P(w), P(t), st) == 0
    return t
def run(name, fn):
    for _ in range(2): assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print(f"{name:14s} {t:7.3f} ms {fl / t / 1e9:6.1f} TF", flush=True)
w0 = bank(0)
run("fwd plain", lambda: lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(w0), None, P(y), 2, ctypes.c_float(0.3), st))
if lib.mpa_conv2d_fold_supported(ctypes.byref(d)):
    w2 = bank(2)
    run("fwd folded", lambda: lib.mpa_conv2d_fwd_folded(ctypes.byref(d), P(x), P(w2), None, P(y), 2, ctypes.c_float(0.3), st))
w1 = bank(1)
run("dgrad " + ("plain" if os.environ.get("MPA_FOLD_OFF") else "folded"), lambda: lib.mpa_conv2d_bwd_data(ctypes.byref(d), P(gy), P(w1), P(dx), st))
