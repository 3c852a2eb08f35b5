"""time conv3's (75,1) filters at T > 75 through the C ABI: python scratch/tall_time.py [B] [T] [Cin] [Cout]"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = int(sys.argv[2]) if len(sys.argv) > 2 else 174
Cin = int(sys.argv[3]) if len(sys.argv) > 3 else 80
Cout = int(sys.argv[4]) if len(sys.argv) > 4 else 50
W = 72
lib = L.load(); dev = torch.device("cuda:0")
d = L.ConvDesc(B, Cin, H, W, Cout, 75, 1, 1, 1, 0, 0)
P = lambda t: ctypes.c_void_p(t.data_ptr()); st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(B, Cin, H, W, device=dev); w = torch.randn(Cout, Cin, 75, 1, device=dev) * 0.03
y = torch.empty(B, Cout, H - 74, W, device=dev); gy = torch.randn_like(y); dx = torch.empty_like(x)
wp = [torch.empty(lib.mpa_conv2d_packed_floats(ctypes.byref(d), m), device=dev) for m in (0, 1)]
for m in (0, 1): assert lib.mpa_conv2d_pack(ctypes.byref(d), m, P(w), P(wp[m]), st) == 0
buf = ctypes.create_string_buffer(512)
fl = 2.0 * B * (H - 74) * W * Cout * Cin * 75
def run(name, fn, mode):
    lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512)
    for _ in range(2): assert fn() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print(f"{name:6s} {t:7.3f} ms {fl / t / 1e9:6.1f} TF  {buf.value.decode()[:100]}", flush=True)
run("fwd", lambda: lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp[0]), None, P(y), 2, ctypes.c_float(0.3), st), 0)
run("dgrad", lambda: lib.mpa_conv2d_bwd_data(ctypes.byref(d), P(gy), P(wp[1]), P(dx), st), 1)
dw = torch.empty_like(w); db = torch.empty(Cout, device=dev)
n = lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d)); ws = torch.empty(n // 4, device=dev)
run("wgrad", lambda: lib.mpa_conv2d_bwd_weight(ctypes.byref(d), P(x), P(gy), P(dw), P(db), P(ws), n, st), 2)
