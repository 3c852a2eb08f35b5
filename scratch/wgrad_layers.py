"""fwd / dgrad / wgrad time of a list of layers (best of 3), for A/B tests of planner changes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import ops
dev = torch.device("cuda:0")
LAYERS = [(256,128,75,216,80,3,3,1,3,1,0),(256,128,75,216,200,3,3,1,3,1,0),(128,128,75,216,180,3,3,1,3,1,0),(128,128,75,216,150,3,3,1,3,1,0),
          (256,64,37,108,32,9,9,1,1,4,4),(256,32,37,108,16,9,9,1,1,4,4),(256,64,18,54,64,9,9,1,1,4,4),(256,32,18,54,64,9,9,1,1,4,4),
          (256,128,18,54,64,5,5,1,1,2,2),(256,64,18,54,32,5,5,1,1,2,2),(256,64,9,27,128,5,5,1,1,2,2),(256,128,9,27,128,5,5,1,1,2,2),
          (256,256,9,27,128,3,3,1,1,1,1),(256,128,9,27,64,3,3,1,1,1,1),(256,128,4,13,128,3,3,1,1,1,1),
          (32,128,75,216,80,3,3,1,3,1,0),(32,64,37,108,32,9,9,1,1,4,4),(32,128,18,54,64,5,5,1,1,2,2),(64,70,75,216,70,3,3,1,3,1,0)]
for (B,Cin,H,W,Cout,kh,kw,sh,sw,ph,pw) in LAYERS:
    x = torch.randn(B,Cin,H,W, device=dev).requires_grad_(True)
    w = (torch.randn(Cout,Cin,kh,kw, device=dev)*0.02).requires_grad_(True)
    b = torch.zeros(Cout, device=dev, requires_grad=True)
    keys = []
    ops.set_kernel_probe(lambda k, kind: (keys.append(kind) or True))
    for _ in range(3):
        y = ops.conv2d(x, w, b, (sh,sw), (ph,pw)); y.backward(torch.ones_like(y))
    ms = ops.probe_results_ms(); ops.set_kernel_probe(None)
    fl = 2.0*y.numel()*Cin*kh*kw
    best = {}
    for k, t in zip(keys, ms): best[k] = min(best.get(k, 1e9), t)
    print(f"B{B} {Cin}->{Cout} {kh}x{kw} @{H}x{W} s{sw}: " + "  ".join(f"{k} {t:.3f}ms {fl/t/1e9:.0f}TF" for k,t in best.items()), flush=True)
