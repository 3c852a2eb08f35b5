"""BASELINE.json configurations at (reduced) training batch sizes on the GPU: every paper-size model runs a full
train step (forward, loss, backward, fused AdamW), stays finite, reduces the loss on a repeated batch, and its
eval-mode forward agrees with the CPU oracle.  Runs on the GPU box only (-m gpu)."""
import numpy as np
import pytest
import torch

from helpers import oracle_forward
from multipitch_architectures_amd import nn_models, ops
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss, PolyphonyLoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.synth import synth_batch

pytestmark = pytest.mark.gpu

CASES = [("CNN:XS", 8), ("DRCNN:L", 4), ("Unet:L", 8), ("SAUnet:L", 25), ("SAUSnet:L", 6), ("BLUnet:XXL", 8),
         ("PUnet:XL", 4), ("BLUnet:L", 5)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.mark.parametrize("name,B", CASES, ids=[f"{n}-B{b}" for n, b in CASES])
def test_train_steps_default_init(dev, name, B):
    cfg = CONFIGS[name]
    torch.manual_seed(0)
    ops.manual_seed(7)
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"]).to(dev).train()
    is_p = cfg["cls"].endswith("polyphony_classif_softmax")
    crit = PolyphonyLoss() if is_p else BCELoss()
    opt = AdamW(model.parameters(), lr=cfg["lr"])
    x, y = synth_batch(B, 75, seed=99)
    x, y = x.to(dev), y.to(dev)
    losses = []
    for _ in range(4):
        res = model(x)
        loss = crit(res[0], res[1], y) if is_p else crit(res, y)
        opt.zero_grad()
        loss.backward()
        for k, p in model.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
        opt.step()
        losses.append(float(loss.detach()))
    assert np.isfinite(losses).all()
    assert losses[-1] < losses[0], losses          # same batch, 4 AdamW steps, dropout active: the loss must go down
    out = res[0] if is_p else res
    assert out.shape == (B, 1, 1, 72) and float(out.min()) >= 0.0 and float(out.max()) <= 1.0


@pytest.mark.parametrize("name,B", [("DRCNN:L", 2), ("Unet:L", 3), ("BLUnet:XXL", 3), ("PUnet:XL", 2)],
                         ids=lambda v: str(v))
def test_eval_forward_vs_oracle_default_init(dev, name, B):
    """default (PyTorch) initialisation instead of the deterministic fill: a second, independent weight distribution"""
    cfg = CONFIGS[name]
    torch.manual_seed(3)
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model.to(dev).eval()
    x, _ = synth_batch(B, 75, seed=5)
    with torch.no_grad():
        res = model(x.to(dev))
        ref = oracle_forward(name, sd, x, train=False)
    a = (res[0] if isinstance(res, tuple) else res).cpu().numpy()
    b = (ref[0] if isinstance(ref, tuple) else ref).numpy()
    assert np.abs(a - b).max() <= 1e-4


def test_blunet_lstm_depth_above_one_raises_like_the_reference(dev):
    """lstm_depth > 1 constructs but cannot run: `lstm4` is built for x5's C*13 features and is fed x4's C*27
    (unet_cnns.py:1037-1038,1088) -- the reference raises RuntimeError there, and so does the HIP model (never a silent
    reshape)."""
    model = nn_models.u_net_blstm_varlayers(n_chan_input=6, n_chan_layers=[8, 8, 6, 4], n_bins_in=216, n_bins_out=72,
                                            scalefac=16, embed_dim=416, hidden_size=208, lstm_depth=2,
                                            lstm_number=1).to(dev).eval()
    x, _ = synth_batch(1, 75)
    with pytest.raises(RuntimeError, match="embed_dim"):
        model(x.to(dev))
