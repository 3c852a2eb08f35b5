"""per-layer conv timing of one SAUnet:L train step (fwd / dgrad / wgrad), with algorithmic TFLOP/s"""
import sys, os
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
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=0.01); _bce = BCELoss()
from multipitch_architectures_amd.losses import PolyphonyLoss
_pl = PolyphonyLoss()
loss_fn = lambda res, y: _pl(res[0], res[1], y) if isinstance(res, tuple) else _bce(res, y)
T = int(sys.argv[3]) if len(sys.argv) > 3 else 75
x, y = synth_batch(B, T); x, y = x.to(dev), y.to(dev)
def step():
    loss = loss_fn(model(x), y); opt.zero_grad(); loss.backward(); opt.step()
step(); step()
keys = []
ops.set_kernel_probe(lambda k, kind: kind != "gemm" and (keys.append((k, kind)) or True))
step()
ms = ops.probe_results_ms()
ops.set_kernel_probe(None)
rows = []
for (k, kind), t in zip(keys, ms):
    Bq, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw = k
    OH, OW = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    fl = 2.0 * Bq * OH * OW * Cin * Cout * kh * kw
    rows.append((t - fl / 157.3e9, t, fl / t / 1e9, kind, k))
tot = sum(r[1] for r in rows); ideal = sum(r[1] - r[0] for r in rows)
print(f"conv total {tot:.1f} ms, MFMA-ideal {ideal:.1f} ms")
for lost, t, tf, kind, k in sorted(rows, reverse=True)[:45]:
    print(f"lost {lost:6.2f} ms  {t:7.3f} ms {tf:6.1f} TF/s {kind:6s} Cin={k[1]:4d} {k[2]}x{k[3]} Cout={k[4]:4d} k={k[5]}x{k[6]} s={k[7]},{k[8]}")
