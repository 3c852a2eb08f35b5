"""time every pixel-tile shape the forward planner could pick for one layer (MPA_FWD_TILE + mpa_diag_reload):
python scratch/tile_sweep.py B Cin H W Cout k mode(0 fwd, 1 dgrad) [NB PB]"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import _lib as L
lib = L.load()
P = lambda t: ctypes.c_void_p(t.data_ptr())
B, ci, H, W, co, k, mode = [int(v) for v in sys.argv[1:8]]
if len(sys.argv) > 9: os.environ["MPA_FWD_FORCE"] = f"{sys.argv[8]},{sys.argv[9]}"
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
d = L.ConvDesc(B, ci, H, W, co, k, k, 1, 1, k // 2, k // 2)
x = torch.randn(B, ci, H, W, device="cuda"); y = torch.randn(B, co, H, W, device="cuda"); w = torch.randn(co, ci, k, k, device="cuda") * 0.03
def run():
    buf = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512)
    n = lib.mpa_conv2d_packed_floats(ctypes.byref(d), mode)
    if n <= 0: return None, buf.value.decode()
    wp = torch.empty(n, device="cuda")
    if lib.mpa_conv2d_pack(ctypes.byref(d), mode, P(w), P(wp), st) != 0: return None, buf.value.decode()
    f = (lambda: lib.mpa_conv2d_fwd(ctypes.byref(d), P(x), P(wp), None, P(y), 0, ctypes.c_float(0), st)) if mode == 0 else (lambda: lib.mpa_conv2d_bwd_data(ctypes.byref(d), P(y), P(wp), P(x), st))
    if f() != 0: return None, buf.value.decode()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(3): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3, buf.value.decode()
fl = 2.0 * B * H * W * ci * co * k * k
ms, desc = run()
print(f"planner: {ms:7.3f} ms {fl/ms/1e9:6.1f} TF/s  {desc[:120]}", flush=True)
import re
m = re.search(r"fwd<(\d+),(\d+)>", desc); PBv = int(m.group(2)); Pn = PBv * 64
os.environ["MPA_FWD_FORCE"] = f"{m.group(1)},{m.group(2)}"          # the planner's wave tile, every pixel-tile shape
res = []
seen = set()
for TH in range(1, min(H, Pn) + 1):
    twmax = min(W, Pn // TH)
    if twmax < 1: continue
    tx0 = -(-W // twmax)
    for tx in range(tx0, tx0 + 3):
        if tx > W: break
        TW = -(-W // tx)
        if TH * TW < 0.8 * Pn or (TH, TW) in seen: continue
        seen.add((TH, TW))
        os.environ["MPA_FWD_TILE"] = f"{TH},{TW}"; lib.mpa_diag_reload()
        ms, desc = run()
        if ms is None or f"tile={TH}x{TW}" not in desc: continue
        res.append((ms, TH, TW, desc))
res.sort()
for ms, TH, TW, desc in res[:12]:
    print(f"{TH:3d}x{TW:<3d} {ms:7.3f} ms {fl/ms/1e9:6.1f} TF/s  {desc[:110]}")
print(len(res), "tiles timed")
