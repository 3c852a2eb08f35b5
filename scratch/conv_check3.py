import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from multipitch_architectures_amd import ops, _lib as L
lib = L.load()
torch.manual_seed(0)
B = 2
for (ci, co, H, W, k) in [(64, 16, 9, 27, 1), (16, 64, 9, 27, 1), (16, 16, 9, 27, 1), (16, 16, 9, 26, 1), (16, 16, 9, 28, 1), (16, 16, 9, 27, 3)]:
    x = torch.randn(B, ci, H, W); w = torch.randn(co, ci, k, k) / (ci * k * k) ** 0.5
    xr = x.double().requires_grad_(True)
    yr = F.conv2d(xr, w.double(), None, padding=k // 2); gy = torch.randn_like(yr); yr.backward(gy)
    xg, wg = (t.cuda().requires_grad_(True) for t in (x, w))
    y = ops.conv2d(xg, wg, None, (1, 1), (k // 2, k // 2), ops.ACT_NONE, 0.0); y.backward(gy.float().cuda())
    d = L.ConvDesc(B, ci, H, W, co, k, k, 1, 1, k // 2, k // 2)
    b0 = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), 0, b0, 512)
    b1 = ctypes.create_string_buffer(512); lib.mpa_conv2d_describe_plan(ctypes.byref(d), 1, b1, 512)
    ey = (y.cpu().double() - yr).abs(); ex = (xg.grad.cpu().double() - xr.grad).abs()
    print(f"{ci}->{co} {H}x{W} k{k}: y {ey.max().item():.1e} dx {ex.max().item():.1e}")
    print("   fwd  ", b0.value.decode()[:120]); print("   dgrad", b1.value.decode()[:120])
    if ex.max() > 1e-3:
        print("   dx err by column:", [round(v, 3) for v in ex.amax(dim=(0, 1, 2)).tolist()])
        print("   dx err by row:", [round(v, 3) for v in ex.amax(dim=(0, 1, 3)).tolist()])
        print("   dx err by channel:", [round(v, 3) for v in ex.amax(dim=(0, 2, 3)).tolist()][:32])
    if ey.max() > 1e-3:
        print("   y err by column:", [round(v, 3) for v in ey.amax(dim=(0, 1, 2)).tolist()])
