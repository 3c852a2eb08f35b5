import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from multipitch_architectures_amd import ops, _lib as L
lib = L.load()
torch.manual_seed(0)
B = 2
for (ci, co, H, W, k) in [(64, 32, 9, 27, 3), (64, 16, 9, 27, 1), (32, 16, 9, 27, 3), (32, 32, 4, 13, 3), (32, 32, 4, 13, 1), (32, 16, 18, 54, 5), (32, 8, 18, 54, 1)]:
    x = torch.randn(B, ci, H, W); w = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5; b = torch.randn(co) * 0.1
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, b.double(), padding=k // 2); gy = torch.randn_like(yr); yr.backward(gy)
    xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
    y = ops.conv2d(xg, wg, bg, (1, 1), (k // 2, k // 2), ops.ACT_NONE, 0.0); y.backward(gy.float().cuda())
    d = L.ConvDesc(B, ci, H, W, co, k, k, 1, 1, k // 2, k // 2)
    buf = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), 1, buf, 512)
    ey = (y.cpu().double() - yr).abs(); ex = (xg.grad.cpu().double() - xr.grad).abs(); ew = (wg.grad.cpu().double() - wr.grad).abs()
    print(f"{ci}->{co} {H}x{W} k{k}: y {ey.max().item()/yr.abs().max().item():.1e} dx {ex.max().item()/xr.grad.abs().max().item():.1e} dw {ew.max().item()/wr.grad.abs().max().item():.1e} | dgrad plan {buf.value.decode()[:110]}")
