import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd.nn_models.unet_cnns import transformer_enc_layer
from oracle import restate
torch.manual_seed(0)
B, E, H, W, heads, mlp = int(sys.argv[1]), int(sys.argv[2]), 4, 13, int(sys.argv[3]), int(sys.argv[4])
layer = transformer_enc_layer(embed_dim=E, num_heads=heads, mlp_dim=mlp, p_dropout=0.0, pos_encoding=None)
sd = {k: v.clone() for k, v in layer.state_dict().items()}
for k in sd: sd[k] = torch.randn_like(sd[k]) * (0.3 if sd[k].dim() > 1 else 0.1) + (1.0 if "layernorm" in k and k.endswith("weight") else 0.0)
layer.load_state_dict(sd)
x = torch.randn(B, E, H, W)
gy = torch.randn(B, E, H, W)
# oracle in float64
sd64 = {"a." + k: v.double().requires_grad_(True) for k, v in sd.items()}
x64 = x.double().requires_grad_(True)
y64 = restate.transformer_enc_layer(x64, sd64, "a", heads, train=True, p_dropout=0.0, pos_encoding=None)
y64.backward(gy.double())
layer = layer.cuda().train()
xg = x.cuda().requires_grad_(True)
y = layer(xg); y.backward(gy.cuda())
rel = lambda a, b: (a.detach().cpu().double() - b).abs().max().item() / max(b.abs().max().item(), 1e-30)
print("fwd", rel(y, y64.detach()), "dx", rel(xg.grad, x64.grad))
for k, p in layer.named_parameters():
    print(f"{k:28s} {rel(p.grad, sd64['a.' + k].grad):.2e}")
