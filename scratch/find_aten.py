"""Which torch ops of one eager SAUnet:L train step launch ATen kernels (fills, adds, copies)?  Prints op name, count and the
Python stack of the first occurrence -- the list of launches that are not in-tree kernels."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from multipitch_architectures_amd import nn_models
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.synth import synth_batch
name = sys.argv[1] if len(sys.argv) > 1 else "SAUnet:L"
cfg = CONFIGS[name]; dev = torch.device("cuda:0")
model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"]).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-3); crit = BCELoss()
x, y = synth_batch(4, 75); x, y = x.to(dev), y.to(dev)
def step():
    out = model(x)
    loss = crit(out[0] if isinstance(out, tuple) else out, y); opt.zero_grad(); loss.backward(); opt.step()
step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter(); first = {}
for ev in prof.events():
    if ev.name.startswith("aten::") and ev.name not in ("aten::empty", "aten::empty_like", "aten::view", "aten::empty_strided",
                                                          "aten::reshape", "aten::as_strided", "aten::detach", "aten::_unsafe_view",
                                                          "aten::slice", "aten::select", "aten::transpose", "aten::t", "aten::alias",
                                                          "aten::unsqueeze", "aten::squeeze", "aten::expand", "aten::permute",
                                                          "aten::contiguous", "aten::result_type", "aten::to", "aten::lift_fresh",
                                                          "aten::chunk", "aten::split", "aten::narrow", "aten::view_as", "aten::flatten"):
        cnt[ev.name] += 1
        if ev.name not in first:
            first[ev.name] = [s for s in ev.stack if "multipitch" in s or "scratch" in s][:4]
for k, v in cnt.most_common():
    print(v, k, first[k])
