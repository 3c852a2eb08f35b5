"""relative L2 error (whole tensor, vs float64) of the three bf16x3 conv ops and of the exact-fp32 ones, random data"""
import sys, torch, torch.nn.functional as F
sys.path.insert(0, ".")
from multipitch_architectures_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for (B, Cin, H, W, Cout) in [(2, 16, 75, 216, 16), (8, 16, 75, 216, 16), (32, 16, 75, 216, 16), (8, 6, 75, 216, 8)]:
    x = torch.randn(B, Cin, H, W); w = torch.randn(Cout, Cin, 15, 15) / 60; dy = torch.randn(B, Cout, H, W)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, padding=7); yr.backward(dy.double())
    for prec in ("f32", "bf16x3"):
        ops.set_conv_precision(prec)
        xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
        y = ops.conv2d(xd, wd, None, (1, 1), (7, 7)); y.backward(dy.to(dev))
        e = lambda a, b: float((a.detach().cpu().double() - b).norm() / b.norm())
        print(f"B={B} {Cin}->{Cout} {prec:7s}: fwd relL2 {e(y, yr.detach()):.2e}  dgrad {e(xd.grad, xr.grad):.2e}  wgrad {e(wd.grad, wr.grad):.2e}", flush=True)
