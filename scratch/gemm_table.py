"""every GEMM launch of one train step of a configuration, with its time and algorithmic TFLOP/s"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import nn_models, ops
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.synth import synth_batch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "SAUnet:L"]; dev = torch.device("cuda:0")
torch.manual_seed(0)
model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"]).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3); loss_fn = BCELoss()
x, y = synth_batch(B, 75); x, y = x.to(dev), y.to(dev)
def step():
    loss = loss_fn(model(x), y); opt.zero_grad(); loss.backward(); opt.step()
step(); step()
keys = []
ops.set_kernel_probe(lambda k, kind: kind == "gemm" and (keys.append(k) or True))
step()
ms = ops.probe_results_ms(); ops.set_kernel_probe(None)
agg = collections.OrderedDict()
for k, t in zip(keys, ms):
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += t
tot = sum(ms); ideal = sum(2.0 * k[0] * k[1] * k[2] / 157.3e9 for k in keys)
print(f"gemm total {tot:.2f} ms in {len(ms)} launches, MFMA-ideal {ideal:.2f} ms")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, ak, bk, acc, act = k
    print(f"{t:7.3f} ms x{n:3d}  M={M:6d} N={N:5d} K={K:6d} A{'k' if ak else 'm'}-contig B{'k' if bk else 'n'}-contig acc={acc} act={act}  {2.0*M*N*K*n/t/1e9:6.1f} TF/s  C {M*N*4*n/t/1e6:6.0f} GB/s")
