"""full-size training soak: N steps of TrainStep (HIP graph replays) on one fixed synthetic batch per configuration;
the loss must stay finite and fall (the model memorises the batch)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import nn_models, ops
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss, PolyphonyLoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.step import TrainStep
from multipitch_architectures_amd.synth import synth_batch
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for name, B in [("SAUnet:L", 32), ("DRCNN:L", 16), ("Unet:L", 32), ("BLUnet:XXL", 32), ("PUnet:XL", 16), ("CNN:XS", 64),
                ("DCNN:M", 16), ("SAUSnet:L", 16)]:
    cfg = CONFIGS[name]
    torch.manual_seed(0); ops.manual_seed(7)
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"]).to(dev).train()
    is_p = cfg["cls"].endswith("polyphony_classif_softmax")
    lf = PolyphonyLoss() if is_p else BCELoss()
    crit = (lambda r, t: lf(r[0], r[1], t)) if is_p else lf
    opt = AdamW(model.parameters(), lr=cfg["lr"])
    ts = TrainStep(model, crit, opt)
    x, y = synth_batch(B, 75, seed=3); x, y = x.to(dev), y.to(dev)
    t0 = time.time(); losses = []
    for i in range(steps):
        losses.append(ts(x, y))
    losses = [float(l) for l in losses]
    ok = all(l == l and abs(l) < 1e4 for l in losses) and min(losses[-5:]) < losses[0]
    print(f"{name:11s} B={B:3d} graph={ts.graph is not None} loss {losses[0]:.4f} -> {losses[-1]:.4f} (min {min(losses):.4f})"
          f" {'OK' if ok else 'FAIL'}  {time.time() - t0:.1f}s", flush=True)
    del model, opt, ts
    torch.cuda.empty_cache()
