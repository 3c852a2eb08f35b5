"""Fused multi-tensor AdamW -- drop-in for the ``torch.optim.AdamW`` call of the experiment scripts
(exp126a_musicnet_cnn_basic.py:103-108,293): decoupled weight decay, bias correction, no amsgrad.

One kernel launch updates every parameter of a param group (device-resident pointer tables).  It subclasses
``torch.optim.Optimizer`` so ``ReduceLROnPlateau`` (exp126a...py:299-302) can drive ``param_groups[i]['lr']``.

Nothing that changes from step to step is a kernel argument: the learning rate and the step count live in a small
device array (``hyper``), the gradient pointer table is written by a kernel that carries the pointers as arguments.  A whole training step --
forward, backward and this update -- can therefore be captured once in a HIP graph and replayed (``step.TrainStep``).
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
        self.table_epoch = 0       # bumped whenever the device tables are dropped: a captured graph of an older epoch is stale

    def _drop_tables(self):
        """forget the device-side pointer tables / step count: they are rebuilt from ``self.state`` at the next step"""
        if getattr(self, "_tables", None):
            self._tables = {}
        self.table_epoch = getattr(self, "table_epoch", 0) + 1

    def load_state_dict(self, state_dict):
        """as torch.optim.Optimizer.load_state_dict; the step count, learning rate and exp_avg / exp_avg_sq pointers the
        update kernel reads live in device tables seeded from ``self.state`` -- they are rebuilt from the loaded state
        (a graph captured before the load is discarded by ``step.TrainStep``)"""
        super().load_state_dict(state_dict)
        self._drop_tables()

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._drop_tables()

    def invalidate_grad_table(self):
        """A replayed graph re-runs its captured ``mpa_store_ptrs`` and leaves the *graph's* gradient addresses in the
        device table: the host-side cache of what the table holds is stale after every replay (an eager step that
        follows one -- the odd last batch of an epoch -- must upload its own gradient pointers even when the caching
        allocator hands out the same addresses as in the previous eager step)."""
        for tab in self._tables.values():
            tab["gkey"] = None

    def _table(self, gi, plist):
        key = (gi, tuple(p.data_ptr() for p in plist))
        tab = self._tables.get(gi)
        if tab is None or tab["key"] != key:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("AdamW state must exist before a HIP graph is captured: run one eager step first")
            dev = plist[0].device
            for p in plist:
                st = self.state[p]
                if "exp_avg" not in st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            mk = lambda vals: torch.tensor(vals, dtype=torch.int64, device=dev)
            step0 = int(self.state[plist[0]]["step"])          # non-zero after load_state_dict
            tab = {"key": key,
                   "p": mk([p.data_ptr() for p in plist]),
                   "m": mk([self.state[p]["exp_avg"].data_ptr() for p in plist]),
                   "v": mk([self.state[p]["exp_avg_sq"].data_ptr() for p in plist]),
                   "n": mk([p.numel() for p in plist]),
                   "g": torch.zeros(len(plist), dtype=torch.int64, device=dev), "gkey": None,
                   "hyper": torch.tensor([float("nan"), float(step0), 0.0, 0.0], dtype=torch.float64, device=dev),
                   "lr": None,
                   "max": max(p.numel() for p in plist)}
            self._tables[gi] = tab
        return tab

    def sync_hyper(self):
        """Push a changed learning rate (ReduceLROnPlateau writes ``param_groups[i]['lr']``) to the device array the
        update kernel reads.  ``step()`` does it by itself; a replayed graph needs it called before every replay."""
        for gi, group in enumerate(self.param_groups):
            tab = self._tables.get(gi)
            if tab is not None and tab["lr"] != float(group["lr"]):
                if torch.cuda.is_current_stream_capturing():
                    raise RuntimeError("learning rate changed inside a graph capture: call sync_hyper() before it")
                tab["hyper"][0:1].fill_(float(group["lr"]))
                tab["lr"] = float(group["lr"])

    def _upload_grad_table(self, tab, plist):
        """gradient pointers -> device table, by kernels that carry them as arguments (copied at launch: no staging
        buffer to keep alive, no synchronisation, and a captured graph replays them as constants)"""
        gkey = tuple(p.grad.data_ptr() for p in plist)
        if tab["gkey"] == gkey:
            return
        arr = (ctypes.c_void_p * len(gkey))(*gkey)
        rc = L.load().mpa_store_ptrs(ctypes.c_void_p(tab["g"].data_ptr()), arr, len(gkey),
                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        L.check(rc, "mpa_store_ptrs")
        tab["gkey"] = gkey

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
            if tab["lr"] != float(group["lr"]):
                self.sync_hyper()
            self._upload_grad_table(tab, plist)
            vp = lambda t: ctypes.c_void_p(t.data_ptr())
            rc = lib.mpa_adamw_step(vp(tab["p"]), vp(tab["g"]), vp(tab["m"]), vp(tab["v"]), vp(tab["n"]), len(plist),
                                    tab["max"], vp(tab["hyper"]), float(group["betas"][0]), float(group["betas"][1]),
                                    float(group["eps"]), float(group["weight_decay"]),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            L.check(rc, "mpa_adamw_step")
            if not torch.cuda.is_current_stream_capturing():
                self.note_steps(1, groups=[gi])
        ops.bump_param_epoch()
        return loss

    def note_steps(self, n=1, groups=None):
        """host mirror of the device-side step count (``state_dict()`` compatibility): a captured step that was
        replayed n times calls this with n"""
        for gi, group in enumerate(self.param_groups):
            if groups is not None and gi not in groups:
                continue
            for p in group["params"]:
                if p in self.state and "step" in self.state[p]:
                    self.state[p]["step"] += n
