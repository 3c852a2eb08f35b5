"""upsample + pad + concat, forward and backward, at the four decoder levels of SAUnet:L (batch from argv)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
tot_f = tot_b = 0.0
for C1, H1, W1, Cs, Hs, Ws in [(128, 4, 13, 128, 9, 27), (64, 9, 27, 64, 18, 54), (32, 18, 54, 32, 37, 108), (16, 37, 108, 16, 75, 216)]:
    x1 = torch.randn(B, C1, H1, W1, device=dev, requires_grad=True)
    x2 = torch.randn(B, Cs, Hs, Ws, device=dev, requires_grad=True)
    gy = torch.randn(B, Cs + C1, Hs, Ws, device=dev)
    bf = bb = 1e9
    for _ in range(4):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); y = ops.upconcat(x1, x2); e[1].record(); y.backward(gy); e[2].record()
        torch.cuda.synchronize()
        bf = min(bf, e[0].elapsed_time(e[1])); bb = min(bb, e[1].elapsed_time(e[2]))
        x1.grad = None; x2.grad = None
    gb = (B * (Cs + C1) * Hs * Ws + B * Cs * Hs * Ws + B * C1 * H1 * W1) * 4 / 1e9
    print(f"{C1:3d}x{H1}x{W1} -> {Cs + C1}x{Hs}x{Ws}: fwd {bf*1e3:7.1f} us ({gb / bf:5.2f} TB/s)  bwd {bb*1e3:7.1f} us")
    tot_f += bf; tot_b += bb
print(f"total fwd {tot_f:.3f} ms  bwd {tot_b:.3f} ms")
