import sys; sys.path.insert(0, "/root/repo")
import torch, torch.nn.functional as F
from multipitch_architectures_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
for (B, Cin, H, W, Cout) in [(2, 8, 37, 216, 6), (2, 20, 37, 216, 20), (1, 16, 12, 30, 24)]:
    x = torch.randn(B, Cin, H, W, generator=g); w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1; b = torch.randn(Cout, generator=g) * 0.1
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride=(3, 3)); gy = torch.randn(yr.shape, generator=g); yr.backward(gy.double())
    xg, wg, bg = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    try:
        y = ops.conv2d(xg, wg, bg, (3, 3), (0, 0))
        print("fwd err", float((y.cpu().double() - yr).abs().max()))
        y.backward(gy.to(dev))
        print("dx err", float((xg.grad.cpu().double() - xr.grad).abs().max()), "dw err", float((wg.grad.cpu().double() - wr.grad).abs().max()))
    except Exception as e:
        print("EXC", type(e).__name__, str(e)[:200])
