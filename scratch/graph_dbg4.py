import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import nn_models, ops
from multipitch_architectures_amd.step import TrainStep
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.synth import det_fill, synth_batch
dev = torch.device("cuda:0")
unrelated = torch.zeros(8, device=dev, dtype=torch.float64)

def run(use_graph, mode):
    cfg = CONFIGS["tiny:CNN"]
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    model.load_state_dict(det_fill(model.state_dict()))
    model.to(dev).train()
    ops.manual_seed(77)
    opt = AdamW(model.parameters(), lr=1e-3)
    ts = TrainStep(model, BCELoss(), opt, use_graph=use_graph)
    out = []
    for i in range(6):
        seed = 100 if "samedata" in mode else 100 + (i % 2)
        x, y = synth_batch(6, 75, seed=seed)
        xd, yd = x.to(dev), y.to(dev)
        if i == 3 and "poke_before_copy" in mode:
            unrelated.fill_(3.0)
        if use_graph and ts.graph is not None and "manualcopy" in mode:
            ts._x.copy_(xd); ts._y.copy_(yd)
            if i == 3 and "poke_after_copy" in mode:
                unrelated.fill_(3.0)
            xd, yd = ts._x, ts._y
        elif i == 3 and "poke_after_copy" in mode:
            unrelated.fill_(3.0)      # (before TrainStep's own copies)
        if i == 3 and "hostsync" in mode:
            torch.cuda.synchronize()
        out.append(float(ts(xd, yd)))
    return out

for mode in sys.argv[1:]:
    e = run(False, mode); g = run(True, mode)
    print(mode, ["%.5f" % v for v in e], ["%.5f" % v for v in g], flush=True)
