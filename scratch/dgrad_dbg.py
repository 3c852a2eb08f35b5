import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import ops
dev = torch.device("cuda:0")
B,Cin,H,W,Cout,kh,kw,sh,sw,ph,pw = 256,128,75,216,80,3,3,1,3,1,0
x = torch.randn(B,Cin,H,W, device=dev).requires_grad_(True)
w = (torch.randn(Cout,Cin,kh,kw, device=dev)*0.02).requires_grad_(True)
keys = []
ops.set_kernel_probe(lambda k, kind: (keys.append(kind) or True))
for _ in range(3):
    y = ops.conv2d(x, w, None, (sh,sw), (ph,pw)); y.backward(torch.ones_like(y))
ms = ops.probe_results_ms()
best = {}
for k, t in zip(keys, ms): best[k] = min(best.get(k, 1e9), t)
print(os.environ.get("MPA_DEBUG_FWD", "0"), best)
