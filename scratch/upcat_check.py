import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from multipitch_architectures_amd import ops
torch.manual_seed(0)
for (B, C1, H1, W1, Cs, Hs, Ws) in [(2, 8, 4, 13, 8, 9, 27), (2, 8, 9, 27, 4, 18, 54), (2, 4, 18, 54, 4, 37, 108), (2, 4, 37, 108, 4, 75, 216), (2, 128, 4, 13, 128, 9, 27)]:
    x1, x2 = torch.randn(B, C1, H1, W1), torch.randn(B, Cs, Hs, Ws)
    for dt in (torch.float32, torch.float64):
        a, b = x1.to(dt).clone().requires_grad_(True), x2.to(dt).clone().requires_grad_(True)
        up = F.interpolate(a, scale_factor=2, mode="bilinear", align_corners=True)
        dY, dX = Hs - up.shape[2], Ws - up.shape[3]
        ref = torch.cat([b, F.pad(up, [dX // 2, dX - dX // 2, dY // 2, dY - dY // 2])], dim=1)
        gy = torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)).to(dt)
        ref.backward(gy)
        if dt == torch.float32: a32 = a.grad.clone()
        else: a64 = a.grad.clone()
    ag, bg = x1.cuda().requires_grad_(True), x2.cuda().requires_grad_(True)
    out = ops.upconcat(ag, bg)
    out.backward(torch.randn(ref.shape, generator=torch.Generator().manual_seed(3)).cuda())
    sc = a64.abs().max().item()
    print((B, C1, H1, W1, Cs, Hs, Ws), "ours vs f64", (ag.grad.cpu().double() - a64).abs().max().item() / sc, "aten f32 vs f64", (a32.double() - a64).abs().max().item() / sc)
