"""ORACLE tooling -- goldens for the U-Net variants the reference exports but no experiment uses (round 3:
simple_u_net, simple_u_net_selfattn, simple_u_net_sixselfattn, simple_u_net_doubleselfattn_alllayers / _varlayers,
simple_u_net_polyphony_classif, simple_u_net_doubleselfattn_polyphony / _polyphony_classif).  Build container only.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_goldens_variants.py

Imports ``libdl.nn_models`` from /root/reference, builds each class at a tiny size with the deterministic weight fill,
and stores data only (tests/golden/xcls-<class>.npz): the state_dict schema, the evaluation output(s) on the seeded
synthetic batch, and for one train-mode step with every dropout p forced to 0 the loss and 16 gradient samples + norm per
parameter.  These files are not picked up by the oracle-pinning tests (no CPU restatement exists for the variants): the
HIP classes are compared with the reference's numbers directly (tests/test_gpu_variants.py).
"""
import json
import os
import sys
import unittest.mock

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
sys.dont_write_bytecode = True

from multipitch_architectures_amd.configs import VARIANT_CONFIGS  # noqa: E402
from multipitch_architectures_amd.synth import det_fill, synth_batch  # noqa: E402
import libdl.nn_models as ref_models  # noqa: E402

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
_real_zeros = torch.zeros


def _zeros_cpu(*a, **kw):
    if "device" in kw and "cuda" in str(kw["device"]):
        kw["device"] = "cpu"
    return _real_zeros(*a, **kw)


def sample_idx(n, k=16):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def logsoftmax_weights(shape, dtype):
    """fixed pattern for the linear loss of the log-softmax classes (their outputs are log-probabilities: no BCE)"""
    n = int(np.prod(shape))
    return torch.cos(torch.arange(n, dtype=torch.float64) * 0.37).reshape(shape).to(dtype)


def variant_loss(name, res, y):
    if "logsoftmax" in name:
        return (res * logsoftmax_weights(tuple(res.shape), res.dtype)).mean()
    two = isinstance(res, tuple)
    if not two and res.dim() == 5:           # simple_u_net_doubleselfattn_transenc returns (B,1,1,T-74,72)
        res = res.squeeze(1)
    loss = torch.nn.BCELoss()(res[0] if two else res, y)
    if two:      # a loss that reaches the polyphony head whatever its width: mean of its (ReLU) output
        loss = loss + res[1].mean() / 25.0
    return loss


# the exported *layer* transformer_temporal_enc_layer on its own (B, C, T', F') interface
LAYER_CASES = {"transformer_temporal_enc_layer": (dict(embed_dim=24, num_heads=4, mlp_dim=16, p_dropout=0.0,
                                                       pos_encoding="sinusoidal"), (3, 4, 10, 6))}


def layer_goldens():
    for name, (kwargs, shape) in LAYER_CASES.items():
        with unittest.mock.patch("torch.zeros", _zeros_cpu):
            layer = getattr(ref_models, name)(**kwargs).double()
        layer.load_state_dict(det_fill(layer.state_dict()))
        layer.pe = layer.pe.double()
        x = torch.randn(shape, generator=torch.Generator().manual_seed(7), dtype=torch.float64, requires_grad=True)
        layer.train()                      # (p_dropout = 0)
        y = layer(x)
        w = logsoftmax_weights(tuple(y.shape), y.dtype)
        (y * w).sum().backward()
        out = {"schema": np.array(json.dumps({k: list(v.shape) for k, v in layer.state_dict().items()})),
               "kwargs": np.array(json.dumps(kwargs)), "x": x.detach().numpy(), "y": y.detach().numpy(),
               "dx": x.grad.numpy()}
        for k, p in layer.named_parameters():
            out[f"grad.{k}"] = p.grad.numpy()
        np.savez_compressed(os.path.join(GOLDEN_DIR, f"xlayer-{name}.npz"), **out)
        print(f"{name:48s} y[min,max]=({out['y'].min():.4f},{out['y'].max():.4f})", flush=True)


def main():
    only = set(sys.argv[1:])
    if not only or only & set(LAYER_CASES):
        layer_goldens()
    for name, kwargs in VARIANT_CONFIGS.items():
        if only and name not in only:
            continue
        B, T = 3, 75
        with unittest.mock.patch("torch.zeros", _zeros_cpu):
            model = getattr(ref_models, name)(**kwargs)
        model.load_state_dict(det_fill(model.state_dict()))
        import inspect
        sig = [(q.name, q.default) for q in list(inspect.signature(getattr(ref_models, name).__init__).parameters.values())[1:]]
        out = {"schema": np.array(json.dumps({k: list(v.shape) for k, v in model.state_dict().items()})),
               "signature": np.array(json.dumps(sig)),
               "B": np.array(B), "T": np.array(T)}
        x, y = synth_batch(B, T, seed=1234)
        model.eval()
        with torch.no_grad():
            res = model(x)
        two = isinstance(res, tuple)
        out["y"] = (res[0] if two else res).numpy()
        if two:
            out["n_pred"] = res[1].numpy()
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        model.train()
        res = model(x)
        loss = variant_loss(name, res, y)
        loss.backward()
        out["train.loss"] = np.array(loss.item())
        out["train.y"] = (res[0] if two else res).detach().numpy()
        for k, p in model.named_parameters():
            if p.grad is None:           # constructed but unused (attention_time3..6 of the _transenc class)
                continue
            out[f"grad.{k}.norm"] = np.array(p.grad.double().norm().item())
            out[f"grad.{k}.absmax"] = np.array(p.grad.abs().max().item())
            out[f"grad.{k}.samples"] = p.grad.numpy().ravel()[sample_idx(p.grad.numel())]
        # the same step with the reference cast to float64: the yardstick for fp32 rounding noise under train-mode BatchNorm
        with unittest.mock.patch("torch.zeros", _zeros_cpu):
            m64 = getattr(ref_models, name)(**kwargs)
        m64.load_state_dict(det_fill(m64.state_dict()))
        m64 = m64.double()
        for m in m64.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
            if hasattr(m, "pe") and isinstance(m.pe, torch.Tensor) and not isinstance(m.pe, torch.nn.Parameter):
                m.pe = m.pe.double()
        m64.train()
        r64 = m64(x.double())
        l64 = variant_loss(name, r64, y.double())
        l64.backward()
        out["train.loss64"] = np.array(l64.item())
        for k, p in m64.named_parameters():
            if p.grad is None:
                continue
            out[f"grad64.{k}.samples"] = p.grad.numpy().ravel()[sample_idx(p.grad.numel())]
            out[f"grad64.{k}.absmax"] = np.array(p.grad.abs().max().item())
        np.savez_compressed(os.path.join(GOLDEN_DIR, f"xcls-{name}.npz"), **out)
        print(f"{name:48s} y[min,max]=({out['y'].min():.4f},{out['y'].max():.4f}) loss {loss.item():.5f}", flush=True)


if __name__ == "__main__":
    main()
