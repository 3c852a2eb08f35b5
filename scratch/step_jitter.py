import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import nn_models, ops
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.synth import synth_batch
B = int(sys.argv[1]); K = int(sys.argv[2]); sync_each = int(sys.argv[3])
cfg = CONFIGS["SAUnet:L"]
torch.manual_seed(0)
dev = torch.device("cuda:0")
model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"]).to(dev).train()
loss_fn = BCELoss(); opt = AdamW(model.parameters(), lr=1e-3, weight_decay=0.01)
x, y = synth_batch(B, 75, seed=1234); x, y = x.to(dev), y.to(dev)
def step():
    loss = loss_fn(model(x), y); opt.zero_grad(); loss.backward(); opt.step(); return loss
for _ in range(3): step()
import gc
if os.environ.get("NOGC"): gc.collect(); gc.freeze(); gc.disable()
torch.cuda.synchronize()
host = []; tot0 = time.perf_counter()
for i in range(K):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter()
    if sync_each: torch.cuda.synchronize()
    host.append((t1 - t0) * 1e3)
torch.cuda.synchronize()
print("B", B, "ms/step", (time.perf_counter() - tot0) / K * 1e3, "host ms per step:", " ".join(f"{h:.1f}" for h in host))
