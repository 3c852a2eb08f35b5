import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from multipitch_architectures_amd import ops, _lib as L
lib = L.load()
def plan(d, mode):
    buf = ctypes.create_string_buffer(1024); lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 1024); return buf.value.decode()
torch.manual_seed(0)
B, Cin, H, W, Cout, k = 1, 16, 75, 216, 128, 15
x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, k, k) / (Cin * k * k) ** 0.5; b = torch.randn(Cout) * 0.1
xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
yr = F.conv2d(xr, wr, b.double(), padding=7); gy = torch.randn_like(yr); yr.backward(gy)
d = L.ConvDesc(B, Cin, H, W, Cout, k, k, 1, 1, 7, 7)
print(plan(d, 0)); print(plan(d, 1))
xg, wg, bg = (t.cuda().requires_grad_(True) for t in (x, w, b))
y = ops.conv2d(xg, wg, bg, (1, 1), (7, 7), ops.ACT_NONE, 0.0); y.backward(gy.float().cuda())
print("y err", (y.cpu().double() - yr).abs().max().item(), "dx err", (xg.grad.cpu().double() - xr.grad).abs().max().item(), "scale", xr.grad.abs().max().item())
e = (xg.grad.cpu().double() - xr.grad).abs()[0]
print("err by channel", e.amax(dim=(1, 2)))
print("err rows", e.amax(dim=(0, 2))[:80])
