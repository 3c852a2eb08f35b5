"""Fused multi-tensor AdamW -- drop-in for the ``torch.optim.AdamW`` call of the experiment scripts
(exp126a_musicnet_cnn_basic.py:103-108,293): decoupled weight decay, bias correction, no amsgrad.

One kernel launch updates every parameter of a param group (device-resident pointer tables).  It subclasses
``torch.optim.Optimizer`` so ``ReduceLROnPlateau`` (exp126a...py:299-302) can drive ``param_groups[i]['lr']``.
"""
import ctypes

import torch

from . import _lib as L
from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("amsgrad=True is not used by the reference and is not built")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid AdamW hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False))
        self._tables = {}

    def _table(self, gi, plist):
        key = (gi, tuple(p.data_ptr() for p in plist))
        tab = self._tables.get(gi)
        if tab is None or tab["key"] != key:
            dev = plist[0].device
            for p in plist:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            mk = lambda vals: torch.tensor(vals, dtype=torch.int64, device=dev)
            tab = {"key": key,
                   "p": mk([p.data_ptr() for p in plist]),
                   "m": mk([self.state[p]["exp_avg"].data_ptr() for p in plist]),
                   "v": mk([self.state[p]["exp_avg_sq"].data_ptr() for p in plist]),
                   "n": mk([p.numel() for p in plist]),
                   "max": max(p.numel() for p in plist)}
            self._tables[gi] = tab
        return tab

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = L.load()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            for p in plist:
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous() or not p.grad.is_contiguous():
                    raise RuntimeError("multipitch_architectures_amd.optim.AdamW needs contiguous fp32 HIP parameters "
                                       "(no CPU fallback)")
            # all parameters of a group share the step count
            tab = self._table(gi, plist)
            step = self.state[plist[0]]["step"] + 1
            for p in plist:
                self.state[p]["step"] = step
            gptr = torch.tensor([p.grad.data_ptr() for p in plist], dtype=torch.int64, device=plist[0].device)
            vp = lambda t: ctypes.c_void_p(t.data_ptr())
            rc = lib.mpa_adamw_step(vp(tab["p"]), vp(gptr), vp(tab["m"]), vp(tab["v"]), vp(tab["n"]), len(plist),
                                    tab["max"], float(group["lr"]), float(group["betas"][0]), float(group["betas"][1]),
                                    float(group["eps"]), float(group["weight_decay"]), int(step),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            L.check(rc, "mpa_adamw_step")
        ops.bump_param_epoch()
        return loss
