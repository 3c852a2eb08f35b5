"""GPU: the experiment runner (data pipeline -> model -> loss -> optimiser -> early stopping -> evaluation) end to end on
synthetic recordings: the loss goes down, the log lines have the scripts' format, the measures come out finite."""
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_runner_trains_and_tests_on_synthetic_recordings(tmp_path):
    import importlib.util
    import os
    from multipitch_architectures_amd import experiment
    spec = importlib.util.spec_from_file_location(
        "run_experiment", os.path.join(os.path.dirname(os.path.dirname(__file__)), "experiments", "run_experiment.py"))
    run = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run)
    torch.manual_seed(0)
    model, criterion, cfg = experiment.build("tiny:CNN")
    train_files = [run.synthetic_recording(6000, 100 + k) for k in range(3)]
    val_files, test_files = [run.synthetic_recording(1500, 7)], [run.synthetic_recording(400, 8)]
    lines = []
    ckpt = str(tmp_path / "best.pt")
    hist = experiment.train(model, criterion, train_files, val_files, lr=cfg["lr"], max_epochs=4, path_trained_model=ckpt,
                            log=lines.append)
    assert len(hist) == 4 and hist[-1][0] < 0.8 * hist[0][0] and hist[-1][1] < hist[0][1]     # both losses decrease
    assert any(re.fullmatch(r"Epoch #\d+ finished\. Train Loss: \d\.\d{4}, Val Loss: \d\.\d{4} with lr: \d\.\d{5}", l)
               for l in lines)
    assert "  .... model of epoch 0 saved." in lines
    model.load_state_dict(torch.load(ckpt))
    mean, framewise = experiment.test(model, test_files, ["synthetic"], log=lines.append)
    assert set(mean) == set(experiment.MEASURES) and all(np.isfinite(v) for v in mean.values())
    assert 0.0 < mean["roc_auc_measure"] <= 1.0
    assert any(l.startswith("Mean f_measure:   ") for l in lines) and any(l.startswith("Framewise ") for l in lines)


def test_exp2_variant_uses_stride_20_and_caps_the_epoch(monkeypatch):
    """Exp2 ('moresamples' / RETRAIN scripts): stride 20 for train and val, and an epoch ends after the batch that makes
    n_batches > cap (RETRAIN_exp180d...py:38-50, 337-338) -- with cap 3 that is 4 optimiser steps per epoch."""
    from multipitch_architectures_amd import experiment
    from multipitch_architectures_amd.synth import synth_file
    assert experiment.VARIANTS["Exp2"] == {"stride": 20, "max_batches": 3800}
    assert experiment.VARIANTS["Exp1"] == {"stride": 50, "max_batches": None}
    assert experiment.VARIANTS["Exp3"] == {"stride": 10, "max_batches": None}
    assert experiment.VARIANTS["Exp4"] == {"stride": 35, "max_batches": 3800}
    assert experiment.EXP4_STRIDES["Schubert_Winterreise"] == (6, 4) and experiment.EXP4_STRIDES["PHENICX-Anechoic"] == (2, None)
    torch.manual_seed(0)
    model, criterion, cfg = experiment.build("tiny:CNN")
    seen, steps = [], []
    real_ds = experiment.dataset_context
    monkeypatch.setattr(experiment, "dataset_context", lambda i, t, p, **kw: (seen.append(p["stride"]), real_ds(i, t, p, **kw))[1])
    real_step = experiment.AdamW.step
    monkeypatch.setattr(experiment.AdamW, "step", lambda self, *a, **k: (steps.append(1), real_step(self, *a, **k))[1])
    files = [synth_file(frames=3000, seed=1)]
    hist = experiment.train(model, criterion, files, files, max_epochs=2, variant="Exp2", max_batches=3,
                            log=lambda *_: None, use_graph=False)       # kernel-by-kernel: every step calls AdamW.step
    assert seen == [20, 20] and len(hist) == 2
    # Exp4: a stride per recording (third tuple element) overrides the variant's
    seen.clear()
    inp, tgt = files[0]
    experiment.train(model, criterion, [(inp, tgt, 6), (inp, tgt)], [(inp, tgt, 4)], max_epochs=1, variant="Exp4",
                     max_batches=1, log=lambda *_: None, use_graph=False)
    assert seen == [6, 35, 4]
    assert len(steps) == 2 * 4 + 2                       # 147 patches at stride 20 = 6 batches of 25 when uncapped


@pytest.mark.parametrize("name", ["tiny:CNN", "tiny:DRCNN", "tiny:Unet"])
def test_segment_wise_inference(name):
    """`predict_file(segment=L)` feeds (6, L+74, 216) windows (hcqt_datasets.py:144-289 shape) instead of
    one 75-frame patch per frame (exp126a...py:427-443).  Exact properties: segment=1 *is* the per-patch loop; window k
    of the result is model(window k) -- including the shorter last window; the result has one row per frame.  It is an
    approximation of the per-patch predictions (different zero-padding borders), so that deviation is only bounded
    loosely here and reported by bench.py."""
    from multipitch_architectures_amd import experiment, nn_models
    from multipitch_architectures_amd.configs import CONFIGS
    from multipitch_architectures_amd.synth import det_fill, synth_file
    dev = torch.device("cuda:0")
    cfg = CONFIGS[name]
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    model.load_state_dict(det_fill(model.state_dict()))
    model.to(dev).eval()
    inputs, targets = synth_file(frames=230, seed=5)
    ref = experiment.predict_file(model, inputs, targets).cpu()                    # the reference's loop
    assert tuple(ref.shape) == (230, 72)
    one = experiment.predict_file(model, inputs, targets, segment=1).cpu()
    assert float((one - ref).abs().max()) <= 1e-5
    L = 100
    seg = experiment.predict_file(model, inputs, targets, segment=L).cpu()         # windows 100, 100, 30
    assert tuple(seg.shape) == (230, 72)
    padded = np.pad(inputs, ((0, 0), (37, 38), (0, 0)))
    with torch.no_grad():
        for k, (s, n) in enumerate([(0, 100), (100, 100), (200, 30)]):
            win = torch.from_numpy(np.log(1.0 + 10.0 * padded[None, :, s:s + n + 74, :]).astype(np.float32)).to(dev)
            want = model(win).cpu()[0, 0]
            assert float((seg[s:s + n] - want).abs().max()) <= 1e-5, k   # logf vs np.log, batch-dependent plans
    assert float((seg - ref).abs().max()) < 0.5 and torch.isfinite(seg).all()
    # the runner's test() accepts the switch and produces the same measures layout
    mean, _ = experiment.test(model, [(inputs, targets)], ["f"], log=lambda *_: None, segment=L)
    assert set(mean) == set(experiment.MEASURES)


@pytest.mark.parametrize("name", ["tiny:CNN", "tiny:Unet", "tiny:SAUnet"])
def test_file_evaluation_matches_the_oracle_chain(name):
    """The scripts' test flow for one recording (exp126a...py:415-470: pad half a context, one 75-frame patch per frame in
    batches of 50, threshold 0.4, calculate_eval_measures) through the HIP path against the same chain built from the CPU
    oracle (oracle/restate.py on the patches, oracle/restate_metrics.py on its predictions) -- SURVEY 8(d)'s stand-in for
    the F-score on MuN-10, which needs data that is not in the reference: predictions within 1e-4, equal argmax pitch per
    frame, equal thresholded activations, and therefore the same precision / recall / F-measure."""
    from helpers import oracle_forward
    from multipitch_architectures_amd import experiment, nn_models
    from multipitch_architectures_amd.configs import CONFIGS
    from multipitch_architectures_amd.metrics import MEASURES, calculate_eval_measures
    from multipitch_architectures_amd.synth import det_fill, synth_file
    from oracle import restate_metrics
    dev = torch.device("cuda:0")
    cfg = CONFIGS[name]
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    model.load_state_dict(det_fill(model.state_dict()))
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(dev).eval()
    inputs, targets = synth_file(frames=130, seed=9)
    pred = experiment.predict_file(model, inputs, targets).cpu()
    got = calculate_eval_measures(torch.from_numpy(targets), pred, MEASURES, threshold=0.4)
    padded = np.pad(inputs, ((0, 0), (37, 38), (0, 0)))
    X = torch.from_numpy(np.stack([np.log(1.0 + 10.0 * padded[:, i:i + 75]) for i in range(130)]).astype(np.float32))
    with torch.no_grad():                               # batches of 50 consecutive frames, as the reference's DataLoader
        ref = torch.cat([oracle_forward(name, sd, X[i:i + 50], train=False) for i in range(0, 130, 50)])[:, 0, 0]
    assert float((pred - ref).abs().max()) <= 1e-4
    assert torch.equal(pred.argmax(1), ref.argmax(1))
    assert torch.equal(pred >= 0.4, ref >= 0.4)
    want = restate_metrics.all_measures(targets, ref.numpy(), threshold=0.4, use_sklearn=False)
    for m in ("precision", "recall", "f_measure", "binary_accuracy"):
        assert got[m] == pytest.approx(want[m], abs=1e-12), m
    for m in MEASURES:
        assert got[m] == pytest.approx(want[m], rel=2e-4, abs=1e-6), m


@pytest.mark.parametrize("config", ["tiny:SAUnet", "tiny:Unet"])
def test_bf16x3_training_reaches_the_f_measure_of_exact_fp32(config, tmp_path):
    """A/B of a real (small) training run in the two arithmetics (BASELINE.json configs[1] / [4] name reduced precision; the
    mode is opt-in here): the runner's synthetic task, same weights, data and batch order, best-validation checkpoint tested
    with the scripts' flow (threshold 0.4).  The task is small and chaotic: over five dropout seeds the exact-fp32 runs
    themselves scatter by 3-4 pp of F-measure (profiles/r04_bf16x3_train_ab.json, scratch/bfx_train_ab.py: tiny:SAUnet
    0.895 +- 0.039 exact against 0.879 +- 0.011 split-bf16, tiny:Unet 0.877 +- 0.032 against 0.856 +- 0.026 -- the split-bf16
    means are 1.6 / 2.1 pp lower, inside one standard deviation of the exact runs).  The split-bf16 runs are also not run-to-run
    reproducible (channel slices and K splits are added atomically), and single runs stray further than that study's standard
    deviation suggests: repeated in four processes, one tiny:Unet run of this test reached 0.694 where the others gave 0.82-0.86
    (the exact runs: 0.902 / 0.843 every time).  What two seeds per arithmetic can assert without failing one time in four: every
    run learns the task, the best validation loss is in the same league, and
    the split-bf16 mean is not more than 20 pp below the exact one.  The statistics themselves are the profiles/ file."""
    import importlib.util
    import os
    from multipitch_architectures_amd import experiment, ops
    spec = importlib.util.spec_from_file_location(
        "run_experiment", os.path.join(os.path.dirname(os.path.dirname(__file__)), "experiments", "run_experiment.py"))
    run = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run)
    train_files = [run.synthetic_recording(3000, 100 + k) for k in range(12)]
    val_files, test_files = [run.synthetic_recording(3000, 7)], [run.synthetic_recording(1500, 8)]

    def one(precision, seed):
        ops.set_conv_precision(precision)
        try:
            ops.manual_seed(seed)
            torch.manual_seed(0)
            model, criterion, cfg = experiment.build(config)
            ckpt = str(tmp_path / f"{precision}_{seed}.pt")
            hist = experiment.train(model, criterion, train_files, val_files, lr=cfg["lr"], max_epochs=60, path_trained_model=ckpt,
                                    log=lambda *_: None)
            model.load_state_dict(torch.load(ckpt))
            mean, _ = experiment.test(model, test_files, ["synthetic"], log=lambda *_: None)
        finally:
            ops.set_conv_precision("f32")
        return float(mean["f_measure"]), min(h[1] for h in hist)

    exact = [one("f32", s) for s in (1234, 4321)]
    split = [one("bf16x3", s) for s in (1234, 4321)]
    f_exact, f_split = [f for f, _ in exact], [f for f, _ in split]
    assert min(f_exact) > 0.6 and min(f_split) > 0.5, (exact, split)                 # every run learnt the task
    assert np.mean(f_split) >= np.mean(f_exact) - 0.20, (exact, split)
    assert min(v for _, v in split) <= 4.0 * min(v for _, v in exact) + 0.02, (exact, split)   # best validation loss: same league
