import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_gpu_step as T
dev = torch.device("cuda:0")
from multipitch_architectures_amd import optim, step, ops
orig_sync = optim.AdamW.sync_hyper
unrelated = torch.zeros(8, device=dev, dtype=torch.float64)
pinned = torch.zeros(1, dtype=torch.float64).pin_memory()

def make(mode):
    def patched(self):
        changed = any(t is not None and t["lr"] != float(g["lr"]) for t, g in
                      ((self._tables.get(i), g) for i, g in enumerate(self.param_groups)))
        if not changed or torch.cuda.is_current_stream_capturing():
            return orig_sync(self)
        if mode == "after":
            orig_sync(self); torch.cuda.current_stream().synchronize()
        elif mode == "before":
            torch.cuda.current_stream().synchronize(); orig_sync(self)
        elif mode == "memcpy":
            for gi, g in enumerate(self.param_groups):
                t = self._tables[gi]
                pinned[0] = float(g["lr"])
                t["hyper"][0:1].copy_(pinned, non_blocking=True)
                t["lr"] = float(g["lr"])
        elif mode == "dummy_only":       # do NOT change the device lr; launch an unrelated fill instead
            unrelated.fill_(3.0)
            for gi, g in enumerate(self.param_groups):
                if self._tables[gi]["lr"] is None:
                    return orig_sync(self)
                self._tables[gi]["lr"] = float(g["lr"])
        elif mode == "same_value":       # rewrite the OLD learning rate: a fill kernel with no semantic effect
            for gi, g in enumerate(self.param_groups):
                if self._tables[gi]["lr"] is None:
                    return orig_sync(self)
                self._tables[gi]["hyper"][0:1].fill_(self._tables[gi]["lr"])
                self._tables[gi]["lr"] = float(g["lr"])
        else:
            orig_sync(self)
    return patched

for mode in sys.argv[1:]:
    optim.AdamW.sync_hyper = make(mode)
    for rep in range(4):
        le, pe, _, _ = T._run(dev, "tiny:CNN", False, 6, lr_after=(3, 1e-4))
        lg, pg, ts, opt = T._run(dev, "tiny:CNN", True, 6, lr_after=(3, 1e-4))
        print(mode, rep, ["%.6f" % v for v in le], ["%.6f" % v for v in lg])
