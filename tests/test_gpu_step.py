"""The captured training step (step.TrainStep: forward + loss + backward + AdamW as one HIP graph) against the same loop
body launched kernel by kernel.  What must hold for the graph to be a drop-in for the reference's loop
(exp126a_musicnet_cnn_basic.py:318-327): fresh dropout masks at every replay, the AdamW bias correction advancing, a
learning-rate change by ReduceLROnPlateau reaching the replayed kernels, odd-sized batches still working."""
import numpy as np
import pytest
import torch

from multipitch_architectures_amd import nn_models, ops
from multipitch_architectures_amd.configs import CONFIGS
from multipitch_architectures_amd.losses import BCELoss, PolyphonyLoss
from multipitch_architectures_amd.optim import AdamW
from multipitch_architectures_amd.step import TrainStep
from multipitch_architectures_amd.synth import det_fill, synth_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _run(dev, name, use_graph, steps, lr=1e-3, B=6, lr_after=None, batches=None):
    cfg = CONFIGS[name]
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    model.load_state_dict(det_fill(model.state_dict()))
    model.to(dev).train()
    ops.manual_seed(77)
    is_p = cfg["cls"].endswith("polyphony_classif_softmax")
    pl, bce = PolyphonyLoss(), BCELoss()
    crit = (lambda r, t: pl(r[0], r[1], t)) if is_p else bce
    opt = AdamW(model.parameters(), lr=lr)
    ts = TrainStep(model, crit, opt, use_graph=use_graph)
    losses = []
    for i in range(steps):
        if lr_after is not None and i == lr_after[0]:
            opt.param_groups[0]["lr"] = lr_after[1]                    # what ReduceLROnPlateau does
        x, y = synth_batch(B if batches is None else batches[i], 75, seed=100 + (i % 2))
        losses.append(float(ts(x.to(dev), y.to(dev)).detach()))
    return losses, [p.detach().cpu() for p in model.parameters()], ts, opt


def test_graph_replays_match_the_eager_loop(dev):
    """tiny:CNN (dropout, no BatchNorm -- not chaotic): same seeds, 6 steps on alternating batches"""
    le, pe, _, _ = _run(dev, "tiny:CNN", False, 6)
    lg, pg, ts, opt = _run(dev, "tiny:CNN", True, 6)
    assert ts.graph is not None and ts.replays == 5                 # step 0 eager, capture at step 1, replays 1..5
    np.testing.assert_allclose(lg, le, rtol=3e-4)
    for a, b in zip(pg, pe):
        assert float((a - b).abs().max()) <= 3e-4 * max(float(b.abs().max()), 1e-3)
    assert all(st["step"] == 6 for st in opt.state.values())


def test_replays_draw_fresh_dropout_masks_and_follow_the_same_stream(dev):
    """lr = 0: parameters never move, so on a repeated batch the loss changes only through the dropout masks"""
    le, _, _, _ = _run(dev, "tiny:CNN", False, 5, lr=0.0, batches=[4] * 5)
    lg, _, ts, _ = _run(dev, "tiny:CNN", True, 5, lr=0.0, batches=[4] * 5)
    assert ts.replays == 4
    # the two batches alternate: steps 0,2,4 see one batch, 1,3 the other -- same data, different masks
    assert len({round(v, 7) for v in (lg[0], lg[2], lg[4])}) == 3
    np.testing.assert_allclose(lg, le, rtol=1e-6)


def test_learning_rate_change_reaches_the_replayed_update(dev):
    le, pe, _, _ = _run(dev, "tiny:CNN", False, 6, lr_after=(3, 1e-4))
    lg, pg, ts, _ = _run(dev, "tiny:CNN", True, 6, lr_after=(3, 1e-4))
    l_const, _, _, _ = _run(dev, "tiny:CNN", True, 6)
    # (backward-data launches of small grids add channel slices atomically: runs agree to rounding, not bit for bit)
    np.testing.assert_allclose(lg, le, rtol=3e-4)
    assert abs(lg[5] - l_const[5]) > 10 * abs(lg[5] - le[5])         # the change did something, well above that noise
    for a, b in zip(pg, pe):
        assert float((a - b).abs().max()) <= 3e-4 * max(float(b.abs().max()), 1e-3)


def test_kernels_launched_between_replays_do_not_disturb_the_graph(dev):
    """Regression: with hipMemsetAsync *nodes* in the captured step (split-K accumulators, BatchNorm sums) ROCm 7.2
    replayed garbage whenever an ordinary kernel had been launched on the stream since the last host synchronisation
    -- e.g. the patch-extraction kernel of the data loader.  The library zero-fills with kernels now; an unrelated fill
    between two replays must change nothing."""
    other = torch.zeros(1 << 16, device=dev)
    ref, _, _, _ = _run(dev, "tiny:CNN", True, 6)

    def poked():
        cfg = CONFIGS["tiny:CNN"]
        model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
        model.load_state_dict(det_fill(model.state_dict()))
        model.to(dev).train()
        ops.manual_seed(77)
        ts = TrainStep(model, BCELoss(), AdamW(model.parameters(), lr=1e-3))
        out = []
        for i in range(6):
            x, y = synth_batch(6, 75, seed=100 + (i % 2))
            x, y = x.to(dev), y.to(dev)
            if i >= 2:
                other.fill_(float(i))                    # un-synchronised kernel right before the replay
            out.append(float(ts(x, y)))
        return out
    np.testing.assert_allclose(poked(), ref, rtol=3e-4)


def test_odd_batch_runs_eagerly_between_replays(dev):
    sizes = [6, 6, 6, 3, 6]
    le, _, _, _ = _run(dev, "tiny:CNN", False, 5, batches=sizes)
    lg, _, ts, opt = _run(dev, "tiny:CNN", True, 5, batches=sizes)
    assert ts.replays == 3
    np.testing.assert_allclose(lg, le, rtol=3e-4)
    assert all(st["step"] == 5 for st in opt.state.values())


def test_eager_steps_between_replays_use_their_own_gradients(dev):
    """eager -> replay -> eager: a replay re-runs the captured mpa_store_ptrs and leaves the *graph's* gradient addresses
    in the optimizer's device table; the odd batch that follows must upload its own pointers even when the caching
    allocator returns the same p.grad addresses as in the previous odd batch (optim.py:70)"""
    sizes = [6, 6, 3, 6, 3, 6, 3, 6]
    le, pe, _, _ = _run(dev, "tiny:CNN", False, len(sizes), batches=sizes)
    lg, pg, ts, opt = _run(dev, "tiny:CNN", True, len(sizes), batches=sizes)
    assert ts.replays == 4
    np.testing.assert_allclose(lg, le, rtol=3e-4)
    for a, b in zip(pg, pe):
        assert float((a - b).abs().max()) <= 3e-4 * max(float(b.abs().max()), 1e-3)
    assert all(st["step"] == len(sizes) for st in opt.state.values())


def test_optimizer_load_state_dict_after_steps_reaches_the_kernel_and_drops_the_graph(dev):
    """resume from a checkpoint after warm-up steps: step count / exp_avg / exp_avg_sq of the loaded state must be what
    the update kernel uses (they live in device tables), and a graph captured before the load must not be replayed
    (optim.py:42).  Dropout off, so that two runs are comparable step by step."""
    cfg = CONFIGS["tiny:CNN"]

    def fresh():
        model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
        model.load_state_dict(det_fill(model.state_dict()))
        for m in model.modules():
            if hasattr(m, "p") and isinstance(getattr(m, "p"), float):
                m.p = 0.0
        model.to(dev).train()
        opt = AdamW(model.parameters(), lr=1e-3)
        return model, opt, TrainStep(model, BCELoss(), opt)

    def batch(i):
        x, y = synth_batch(4, 75, seed=100 + (i % 2))
        return x.to(dev), y.to(dev)

    # run A: 5 steps, checkpoint after step 3
    model, opt, ts = fresh()
    for i in range(3):
        ts(*batch(i))
    ck_m = {k: v.detach().clone() for k, v in model.state_dict().items()}
    sd = opt.state_dict()
    ck_o = {"state": {k: {n: (t.clone() if torch.is_tensor(t) else t) for n, t in st.items()}
                      for k, st in sd["state"].items()}, "param_groups": sd["param_groups"]}
    for i in range(3, 5):
        ts(*batch(i))
    want = [p.detach().cpu().clone() for p in model.parameters()]
    # run B: another history (2 steps from the initial weights on other batches, graph captured), then the checkpoint
    model2, opt2, ts2 = fresh()
    for i in range(2):
        ts2(*batch(i + 1))
    assert ts2.graph is not None
    model2.load_state_dict(ck_m)
    opt2.load_state_dict(ck_o)
    for i in range(3, 5):
        ts2(*batch(i))
    assert all(st["step"] == 5 for st in opt2.state.values())
    got = [p.detach().cpu() for p in model2.parameters()]
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) <= 3e-4 * max(float(b.abs().max()), 1e-3)


def test_replay_is_refused_after_model_eval(dev):
    """model.eval() between steps: the train-mode graph (dropout, batch statistics) must not be replayed"""
    _, _, ts, _ = _run(dev, "tiny:DRCNN", True, 3, B=4)
    assert ts.graph is not None
    n = ts.replays
    ts.model.eval()
    x, y = synth_batch(4, 75, seed=100)
    ts(x.to(dev), y.to(dev))
    assert ts.replays == n
    ts.model.train()
    ts(x.to(dev), y.to(dev))
    assert ts.replays == n + 1


@pytest.mark.parametrize("name", ["tiny:SAUnet", "tiny:BLUnet", "tiny:PUnet", "tiny:DRCNN"])
def test_every_family_captures_and_trains(dev, name):
    """BatchNorm running statistics, batch-axis attention, the BiLSTM recurrence and the two-headed loss inside a graph"""
    lg, pg, ts, _ = _run(dev, name, True, 6, B=4)
    assert ts.graph is not None and ts.replays == 5
    assert np.isfinite(lg).all() and lg[-1] < lg[0]
    assert all(torch.isfinite(p).all() for p in pg)
    le, _, _, _ = _run(dev, name, False, 2, B=4)
    np.testing.assert_allclose(lg[:2], le, rtol=1e-4)               # step 1 is the first replay


def test_eager_forward_after_replays_sees_the_updated_weights(dev):
    """the packed filter banks are cached per weight version; graph replays change the weights behind PyTorch's back,
    so TrainStep bumps the parameter epoch after every replay -- an eval forward right after training must use the
    current weights (checked against the oracle run on the model's current state_dict)"""
    from helpers import oracle_forward
    _, _, ts, _ = _run(dev, "tiny:Unet", True, 5, B=4)
    model = ts.model
    assert ts.replays == 4
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    model.eval()
    x, _ = synth_batch(3, 75, seed=31)
    with torch.no_grad():
        got = model(x.to(dev)).cpu()
        ref = oracle_forward("tiny:Unet", sd, x, train=False)
    assert float((got - ref).abs().max()) <= 1e-4
    model.train()


def test_evaluation_passes_between_steps_do_not_disturb_the_step_or_see_stale_filters(dev):
    """the step re-packs all its filter banks in one launch at its start (`ops.run_pack_table`); evaluation
    forwards of other shapes between the steps -- also between the first eager step and the capture -- must neither leak
    their banks into the step's table nor read banks packed for older weights"""
    from helpers import oracle_forward
    cfg = CONFIGS["tiny:CNN"]      # (dropout, no BatchNorm: loss sequences are comparable to 2e-5)

    def run(use_graph, with_eval):
        model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
        model.load_state_dict(det_fill(model.state_dict()))
        model.to(dev).train()
        ops.manual_seed(5)
        opt = AdamW(model.parameters(), lr=1e-3)
        ts = TrainStep(model, BCELoss(), opt, use_graph=use_graph)
        xe, _ = synth_batch(3, 90, seed=9)
        losses, evals = [], []
        for i in range(5):
            x, y = synth_batch(4, 75, seed=100 + (i % 2))
            losses.append(float(ts(x.to(dev), y.to(dev))))
            if with_eval:
                model.eval()
                with torch.no_grad():
                    evals.append(model(xe.to(dev)).cpu())
                model.train()
        sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        return losses, evals, sd, xe, ts

    l_ref, _, _, _, _ = run(False, False)
    l_g, ev, sd, xe, ts = run(True, True)
    assert ts.replays == 4
    np.testing.assert_allclose(l_g, l_ref, rtol=2e-5)
    # the last evaluation pass used the weights of step 5, not banks packed for an earlier step
    ref = oracle_forward("tiny:CNN", sd, xe, train=False)
    assert (ev[-1] - ref).abs().max() < 1e-4
    assert (ev[0] - ev[-1]).abs().max() > 1e-6          # ... and the weights did move


def test_bf16x3_filter_banks_survive_evaluation_passes_between_replays(dev):
    """the split-bf16 filter banks a captured graph packs into and reads must stay alive (and in place) whatever
    runs between replays -- an evaluation forward touches only the forward banks, and the backward-data banks of the graph
    used to be handed back to the allocator two weight versions later.  Capture, replay, eval, replay, eval, replay ...
    against the same schedule launched kernel by kernel."""
    def run(use_graph):
        cfg = CONFIGS["tiny:Unet"]
        model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
        model.load_state_dict(det_fill(model.state_dict()))
        model.to(dev).train()
        ops.manual_seed(5)
        opt = AdamW(model.parameters(), lr=2e-4)
        ts = TrainStep(model, BCELoss(), opt, use_graph=use_graph)
        x, y = synth_batch(32, 75, seed=3)             # batch 32: BatchNorm is not chaotic
        x, y = x.to(dev), y.to(dev)
        xe = synth_batch(5, 75, seed=4)[0].to(dev)
        losses, evals = [], []
        for i in range(7):
            losses.append(float(ts(x, y)))
            if i >= 1:
                model.eval()
                with torch.no_grad():
                    evals.append(model(xe).cpu())
                    if i % 2:
                        junk = [torch.empty(1 << 20, device=dev).fill_(float("nan")) for _ in range(8)]   # recycle freed blocks
                        del junk
                model.train()
        return losses, evals, ts

    ops.set_conv_precision("bf16x3")
    try:
        le, ee, _ = run(False)
        lg, eg, ts = run(True)
    finally:
        ops.set_conv_precision("f32")
    assert ts.graph is not None and ts.replays == 6
    assert np.isfinite(lg).all()
    # (rounding drift between the two schedules grows to ~2e-3 of the loss and ~7e-3 of an evaluation output over 7 steps of
    # this BatchNorm net; a bank that was recycled under the graph -- the freed blocks are refilled with NaNs above -- shows
    # as NaN or garbage, not as a third-digit difference)
    np.testing.assert_allclose(lg, le, rtol=1e-2)
    for i, (a, b) in enumerate(zip(eg, ee)):
        assert bool(torch.isfinite(a).all()) and float((a - b).abs().max()) <= 3e-2, (i, float((a - b).abs().max()))
