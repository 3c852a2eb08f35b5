"""input LayerNorm([6,216]) forward / backward at batch B (argv), T = 75"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
x = torch.randn(B, 6, 75, 216, device=dev, requires_grad=True)
w = torch.ones(6, 216, device=dev, requires_grad=True); b = torch.zeros(6, 216, device=dev, requires_grad=True)
gy = torch.randn(B, 6, 75, 216, device=dev)
bf = bb = 1e9
for _ in range(5):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); y = ops.layernorm_cf(x, w, b); e[1].record(); y.backward(gy); e[2].record()
    torch.cuda.synchronize()
    bf = min(bf, e[0].elapsed_time(e[1])); bb = min(bb, e[1].elapsed_time(e[2]))
    x.grad = None; w.grad = None; b.grad = None
print(f"B={B} layernorm_cf fwd {bf*1e3:.1f} us  bwd {bb*1e3:.1f} us")
