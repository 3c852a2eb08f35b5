import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import nn_models, ops, step as stepmod
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.synth import det_fill, synth_batch
dev = torch.device("cuda:0")
unrelated = torch.zeros(8, device=dev, dtype=torch.float64)

class DbgStep(stepmod.TrainStep):
    def eager(self, x, y):
        out = self.model(x)
        self._out = out
        loss = self.criterion(out, y)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        ops.rng_advance()
        return loss

def run(use_graph, poke):
    cfg = CONFIGS["tiny:CNN"]
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    model.load_state_dict(det_fill(model.state_dict()))
    model.to(dev).train()
    ops.manual_seed(77)
    opt = AdamW(model.parameters(), lr=1e-3)
    ts = DbgStep(model, BCELoss(), opt, use_graph=use_graph)
    rows = []
    for i in range(5):
        x, y = synth_batch(6, 75, seed=100 + (i % 2))
        xd, yd = x.to(dev), y.to(dev)
        if poke and i == 3:
            unrelated.fill_(3.0)
        loss = ts(xd, yd)
        torch.cuda.synchronize()
        st = ops._Rng.state[dev].tolist()
        xs = ts._x if (ts.graph is not None and i >= 1 and use_graph) else xd
        rows.append((float(loss.detach()), float(ts._out.double().mean()), float(xs.double().sum()), st[1],
                     float(sum(p.double().abs().sum() for p in model.parameters())),
                     float(sum(p.grad.double().abs().sum() for p in model.parameters()))))
    return rows

for name, g, poke in (("eager", False, False), ("graph", True, False), ("graph+poke", True, True), ("graph+poke", True, True)):
    for r in run(g, poke):
        print(name, " ".join("%.6f" % v for v in r))
