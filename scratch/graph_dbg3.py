import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multipitch_architectures_amd import nn_models, ops, step as stepmod
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.synth import det_fill, synth_batch
dev = torch.device("cuda:0")
mode = sys.argv[1]

class DbgStep(stepmod.TrainStep):
    def eager(self, x, y):
        out = self.model(x)
        if "keepout" in mode:
            self._out = out
        loss = self.criterion(out, y)
        self.opt.zero_grad()
        loss.backward()
        self.opt.step()
        ops.rng_advance()
        return loss

def run(use_graph):
    cfg = CONFIGS["tiny:CNN"]
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    model.load_state_dict(det_fill(model.state_dict()))
    model.to(dev).train()
    ops.manual_seed(77)
    opt = AdamW(model.parameters(), lr=1e-3)
    ts = DbgStep(model, BCELoss(), opt, use_graph=use_graph)
    for i in range(4):
        x, y = synth_batch(6, 75, seed=100 + (i % 2))
        loss = ts(x.to(dev), y.to(dev))
        if "sync" in mode:
            torch.cuda.synchronize()
        if "rngread" in mode:
            ops._Rng.state[dev].tolist()
        if "gradsum" in mode:
            float(sum(p.grad.double().abs().sum() for p in model.parameters()))
        print(mode, use_graph, i, float(loss.detach()), flush=True)

run(False)
run(True)
print(mode, "OK", flush=True)
