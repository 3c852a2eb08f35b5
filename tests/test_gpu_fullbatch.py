"""Every BASELINE.json configuration as a *whole model at its BASELINE batch* through the training step the benchmark
times (step.TrainStep: forward + loss + backward + fused AdamW, captured as one HIP graph) -- what a planner regression
that only bites in the composition at full batch (workspace sizes, pack tables, graph capture at 12 GB) would hit.
Loop body: experiments/Exp1_SectionIV-B/exp126a_musicnet_cnn_basic.py:318-327.

Per configuration (det_fill weights, synthetic HCQT batch, dropout active):
  * 3 captured-graph steps on a repeated batch: finite loss and gradients, loss of step 3 < loss of step 1;
  * the graph's loss sequence == the kernel-by-kernel loop's (same seeds, same dropout stream) to rounding;
  * evaluation forward of 8 samples of the trained state == the CPU oracle <= 1e-4 with equal argmax pitch
    (SAUnet:L: the first 25 samples -- the batch-axis attention makes the output depend on the batch, 25 is the
    reference's training batch and what the oracle can afford).
Runs on the GPU box only (-m gpu)."""
import numpy as np
import pytest
import torch

from helpers import build_model, oracle_forward
from multipitch_architectures_amd import ops
from multipitch_architectures_amd.losses import BCELoss, PolyphonyLoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.step import TrainStep
from multipitch_architectures_amd.synth import synth_batch

pytestmark = pytest.mark.gpu

# (configuration, BASELINE batch, samples of the evaluation check)
# AdamW's first steps move every weight by ~lr whatever its gradient: with the deterministic test fill (weights of one
# scale in every layer) the scripts' 1e-3 / 2e-4 overshoot on a single repeated batch, 2e-5 descends for every family
LR = 2e-5
# (PUnet:XL at 256: the batch SURVEY 8(d)'s FLOP table quotes for it -- 62.4 TFLOP per step)
CASES = [("DRCNN:L", 64, 8), ("Unet:L", 128, 8), ("SAUnet:L", 256, 25), ("BLUnet:XXL", 256, 8), ("PUnet:XL", 256, 8)]


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _criterion(name):
    if name.startswith("PUnet"):
        pl = PolyphonyLoss()
        return lambda r, t: pl(r[0], r[1], t)
    return BCELoss()


def _train(dev, name, B, use_graph, steps=3):
    model = build_model(name, dev).train()
    ops.manual_seed(2026)
    opt = AdamW(model.parameters(), lr=LR, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    ts = TrainStep(model, _criterion(name), opt, use_graph=use_graph)
    x, y = synth_batch(B, 75, seed=77)
    x, y = x.to(dev), y.to(dev)
    losses = [float(ts(x, y)) for _ in range(steps)]
    return model, ts, losses, x


@pytest.fixture(params=["f32", "bf16x3"])
def precision(request):
    """both arithmetics of the convolutions / large GEMMs (ops.set_conv_precision): the exact fp32 default and the opt-in
    split-bf16 mode, each at the BASELINE batch"""
    ops.set_conv_precision(request.param)
    yield request.param
    ops.set_conv_precision("f32")


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"{c[0]}-B{c[1]}")
def test_whole_model_at_baseline_batch(dev, case, precision):
    name, B, n_eval = case
    _, _, l_eager, _ = _train(dev, name, B, use_graph=False)
    torch.cuda.empty_cache()
    model, ts, l_graph, x = _train(dev, name, B, use_graph=True)
    assert ts.graph is not None and ts.replays == 2              # step 1 kernel by kernel, capture at step 2, replay 2 and 3
    assert np.isfinite(l_graph).all() and np.isfinite(l_eager).all()
    assert l_graph[2] < l_graph[0], l_graph                      # a repeated batch: the loss goes down
    # backward-data of small grids adds channel slices atomically: sequences agree to rounding, not bit for bit
    np.testing.assert_allclose(l_graph, l_eager, rtol=2e-3)
    # one more kernel-by-kernel step on the trained state: every gradient is finite
    loss = _criterion(name)(model(x), synth_batch(B, 75, seed=77)[1].to(dev))
    model.zero_grad(set_to_none=True)
    loss.backward()
    for k, p in model.named_parameters():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    # evaluation forward of the trained state against the oracle
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.eval()
    xe = x[:n_eval]
    with torch.no_grad():
        res = model(xe)
        ref = oracle_forward(name, sd, xe.cpu(), train=False)
    a = (res[0] if isinstance(res, tuple) else res).cpu().numpy()
    b = (ref[0] if isinstance(ref, tuple) else ref).numpy()
    assert a.shape == b.shape == (n_eval, 1, 1, 72)
    assert np.abs(a - b).max() <= 1e-4, np.abs(a - b).max()
    for ra, rb in zip(a.reshape(n_eval, 72), b.reshape(n_eval, 72)):
        top = np.sort(rb)[-2:]
        if top[1] - top[0] > 2e-4:
            assert ra.argmax() == rb.argmax()
    if isinstance(res, tuple):
        assert np.abs(res[1].cpu().numpy() - ref[1].numpy()).max() <= 2e-4 * max(1.0, float(ref[1].abs().max()))
    del model, ts
    torch.cuda.empty_cache()
