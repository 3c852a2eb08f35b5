"""per-parameter gradient error of the bf16x3 path against the exact-fp32 HIP path (whole tensors, relative L2 and max/scale)"""
import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import build_model
from multipitch_architectures_amd import ops
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.nn_models.layers import Dropout
from multipitch_architectures_amd.synth import synth_batch
dev = torch.device("cuda:0")
for name, B in [("tiny:CNN", 32), ("tiny:Unet", 32), ("tiny:SAUnet", 32), ("tiny:Unet", 2)]:
    grads = {}
    for prec in ("f32", "bf16x3"):
        ops.set_conv_precision(prec)
        model = build_model(name, dev)
        for m in model.modules():
            if isinstance(m, Dropout): m.p = 0.0
        model.train()
        x, y = synth_batch(B, 75)
        loss = BCELoss()(model(x.to(dev)), y.to(dev))
        loss.backward()
        grads[prec] = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters()}
        grads[prec]["__loss"] = torch.tensor(float(loss), dtype=torch.float64)
    print(f"== {name} B={B}: loss f32 {float(grads['f32']['__loss']):.8f} bf16x3 {float(grads['bf16x3']['__loss']):.8f}")
    tot_n = tot_d = 0.0
    for k in grads["f32"]:
        if k == "__loss": continue
        a, b = grads["f32"][k], grads["bf16x3"][k]
        l2 = float((a - b).norm() / max(float(a.norm()), 1e-30)); mx = float((a - b).abs().max() / max(float(a.abs().max()), 1e-30))
        tot_n += float((a - b).pow(2).sum()); tot_d += float(a.pow(2).sum())
        print(f"   {k:40s} relL2 {l2:.2e}  max/scale {mx:.2e}  |g| {float(a.norm()):.3e}")
    print(f"   whole gradient relL2 {np.sqrt(tot_n / tot_d):.2e}")
