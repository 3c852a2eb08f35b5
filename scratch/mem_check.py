import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import nn_models
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.synth import synth_batch
cfg = CONFIGS["SAUnet:L"]; dev = torch.device("cuda:0")
model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"]).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3, weight_decay=0.01); loss_fn = BCELoss()
x, y = synth_batch(256, 75); x, y = x.to(dev), y.to(dev)
for _ in range(2):
    loss = loss_fn(model(x), y); opt.zero_grad(); loss.backward(); opt.step()
torch.cuda.synchronize()
print("max allocated GB", torch.cuda.max_memory_allocated() / 2**30, "reserved GB", torch.cuda.max_memory_reserved() / 2**30)
