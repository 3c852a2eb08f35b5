#!/usr/bin/env python
"""bench.py -- one "step" = one training step (forward + BCE loss + backward + gradient all-reduce + AdamW) of the
hot path over one batch of synthetic HCQT patches.

    python bench.py --gpus N --steps K --warmup W [--config SAUnet:L] [--global-batch 256] [--frames 75]

Headline workload (BASELINE.json configs[3]): SAUnet:L (exp180d), global batch 256, T=75, fp32, strong scaling
(local batch 256/N), data-parallel with RCCL gradient all-reduce.  Prints ONE JSON line on rank 0.
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="SAUnet:L")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="default: the batch BASELINE.json quotes for the configuration (256 for SAUnet:L)")
    ap.add_argument("--frames", type=int, default=75)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the patch-extraction / evaluation / segment-inference "
                    "probes after the timed region (profiling runs)")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel from the host instead of replaying "
                    "the captured HIP graph of the step")
    ap.add_argument("--conv-precision", choices=("f32", "bf16x3"), default="f32",
                    help="arithmetic of the 15-row convolutions: exact fp32-input MFMA (default, the headline) or the opt-in "
                    "split-bf16 path (hi*hi + hi*lo + lo*hi on bf16 MFMA, fp32 accumulation); the bf16x3 line carries "
                    "dtype 'bf16x3-f32acc' and its roofline against the bf16 MFMA peak / 3")
    ap.add_argument("--sync-bn", action="store_true", help="data-parallel exactness mode: BatchNorm statistics over the global "
                    "batch (all-reduce of the per-channel sums); the step then runs kernel by kernel")
    ap.add_argument("--gather-attention", action="store_true", help="data-parallel exactness mode: the batch-axis attention "
                    "sees the keys / values of every rank")
    ap.add_argument("--dp-rehearsal", action="store_true", help="with --gpus 1: run what a data-parallel rank runs (RCCL "
                    "process group of one rank, gradient hooks, bucketed asynchronous all-reduces, kernel-by-kernel "
                    "step) -- the per-rank cost of the data-parallel machinery on a one-GPU box; not a BASELINE line")
    return ap.parse_args(argv)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: this process becomes the parent of N fresh rank processes (one per
    GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, the same command line) and only waits for them.
    It runs before torch is imported here, so the parent never touches a GPU and no process that has initialised HIP is
    ever re-executed.  Rank 0 inherits stdout and prints the JSON line; any failing rank takes the others down and its
    exit code becomes ours."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            time.sleep(0.2)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {procs.index(p)} exited with {code}; stopping the other ranks",
                          file=sys.stderr, flush=True)
                    break
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except Exception:       # noqa: BLE001 -- a rank that ignores SIGTERM
                p.kill()
    return rc


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _args = parse_args()
    if _args.gpus > 1:
        sys.exit(spawn_ranks(_args.gpus))

import torch  # noqa: E402  (after the spawn decision: the parent of an N>1 run never loads it)

PROBE_STEPS = 3
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X dense fp32-input MFMA peak (MI355X_MICROARCH.md, chip table)
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense bf16 MFMA peak (same table); a split-bf16 product costs three MFMAs


def cpu_baseline(config, frames, sample_batch=8, steps=2):
    """Reference-equivalent CPU path (oracle/restate.py, parity-pinned to the reference) timed on the host cores."""
    from multipitch_architectures_amd import nn_models
    from multipitch_architectures_amd.configs import CONFIGS
    from multipitch_architectures_amd.synth import synth_batch
    from oracle import restate
    cfg = CONFIGS[config]
    torch.manual_seed(0)
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
    sd, names = restate.split_state(model.state_dict())
    x, y = synth_batch(sample_batch, frames, seed=1234)
    state = {}
    fn = restate.MODELS[cfg["cls"]]

    def one():
        res = fn(sd, x, train=True, **cfg["kwargs"])
        loss = restate.punet_loss(res[0], res[1], y) if isinstance(res, tuple) else restate.bce_loss(res, y)
        grads = dict(zip(names, torch.autograd.grad(loss, [sd[k] for k in names])))
        restate.adamw_step({k: sd[k] for k in names}, grads, state, lr=cfg["lr"])

    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    return {"value": sample_batch * (frames - 74) / dt, "unit": "frames/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{steps} train steps of {config} at batch {sample_batch}, T={frames} "
            f"(torch-CPU restatement oracle/restate.py, {dt:.2f} s/step)"}


def patch_extraction_probe(batch, with_cpu):
    """`dataset_context` windows + the training augmentations, one launch per batch from a resident
    recording (not part of the timed train step, whose inputs are already in HBM).  HBM-bound: 4 B read + 4 B written
    per output element."""
    from multipitch_architectures_amd.data_loaders import dataset_context
    from multipitch_architectures_amd.synth import synth_file
    params = {"context": 75, "stride": 50, "compression": 10, "aug:transpsemitones": 5, "aug:randomeq": 20,
              "aug:noisestd": 1e-4, "aug:tuning": True}                    # exp180d...py:38-45
    inputs, targets = synth_file(frames=50 * batch + 100, seed=11)
    ds = dataset_context(inputs, targets, params)
    idx = list(range(batch))
    for _ in range(3):
        ds.batch(idx)
    reps = 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        X, y = ds.batch(idx)                                       # what a training loop calls: host draws + launch
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    # the kernel alone: replay the launch through the C ABI with the already-uploaded index/augmentation tables
    import ctypes
    from multipitch_architectures_amd import _lib as L
    table, aug = X._mpa_keepalive
    vp = lambda a: ctypes.c_void_p(a)
    st = vp(torch.cuda.current_stream().cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for r in range(reps):
        L.check(L.load().mpa_context_batch(ctypes.byref(ds.desc), batch, vp(table.data_ptr()),
                                           vp(table.data_ptr() + 8 * batch), vp(table.data_ptr() + 16 * batch),
                                           vp(aug.data_ptr()), None, None, None, ctypes.c_uint64(r), vp(X.data_ptr()),
                                           vp(y.data_ptr()), st), "mpa_context_batch")
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / reps
    nbytes = 8.0 * X.numel()
    res = {"patches_per_s": batch / wall, "launch_ms": ms, "batch": batch, "bound": "hbm",
           "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
           "frac": nbytes / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_patch": nbytes / batch,
           "note": "launch_ms/achieved: the kernel replayed back-to-back; patches_per_s: wall clock of "
                   "dataset.batch() incl. host-side draws and table upload"}
    if with_cpu:
        from oracle import restate_data as RD                     # checker timed as the CPU baseline, never shipped
        ti, tt = torch.from_numpy(inputs), torch.from_numpy(targets)
        torch.set_num_threads(1)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 3.0:
            RD.context_patch(ti, tt, params, n % batch)
            n += 1
        res["cpu_baseline"] = {"value": n / (time.perf_counter() - t0), "unit": "patches/s", "cores": 1, "kind": "port",
                               "sample": f"{n} patches, one thread (the reference uses 16 DataLoader workers)"}
    return res


def eval_measures_probe(with_cpu, n_frames=20000):
    """the 11 evaluation measures of one recording (about 7.7 min of audio at 43 Hz), predictions already
    on the device.  HBM-bound in principle (8 B per (frame, bin) read once per pass) but launch-latency dominated at
    this size: 1.44 M scores."""
    from multipitch_architectures_amd.metrics import MEASURES, calculate_eval_measures
    from multipitch_architectures_amd.synth import synth_eval_pair
    targ, pred = synth_eval_pair(n_frames=n_frames, seed=3)
    t, p = torch.from_numpy(targ).cuda(), torch.from_numpy(pred).cuda()
    calculate_eval_measures(t, p, MEASURES, threshold=0.4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        got = calculate_eval_measures(t, p, MEASURES, threshold=0.4)       # ends with a device->host copy of 16 doubles
    dt = (time.perf_counter() - t0) / reps
    res = {"frames": n_frames, "bins": 72, "ms_per_recording": dt * 1e3, "frames_per_s": n_frames / dt,
           "f_measure": got["f_measure"]}
    if with_cpu:
        from oracle import restate_metrics as RM                  # checker timed as the CPU baseline, never shipped
        t0 = time.perf_counter()
        RM.all_measures(targ, pred, threshold=0.4)
        res["cpu_baseline"] = {"value": n_frames / (time.perf_counter() - t0), "unit": "frames/s", "cores": 1,
                               "kind": "port", "sample": f"one recording of {n_frames} frames (numpy + scikit-learn)"}
    return res


def hcqt_probe(with_cpu, seconds=30.0):
    """audio -> HCQT (6 harmonics x 216 bins, 43 frames/s: the reference's feature settings,
    01_precompute_features.ipynb cell 5) on the GPU, the signal resident in HBM.  Parity unpinned (librosa absent): see
    DESIGN.md 6b."""
    import numpy as np
    from multipitch_architectures_amd.data_preprocessing import efficient_hcqt_device
    sr = 22050
    rng = np.random.default_rng(5)
    t = np.arange(int(seconds * sr)) / sr
    y = sum(a * np.sin(2 * np.pi * f * t) for f, a in ((220.0, 0.4), (277.2, 0.3), (329.6, 0.3), (440.0, 0.2)))
    y = torch.from_numpy((y + 0.01 * rng.standard_normal(len(t))).astype(np.float32)).cuda()
    kw = dict(fs=sr, fs_hcqt_target=50, bins_per_octave=36, num_octaves=6, num_harmonics=5, num_subharmonics=1)
    efficient_hcqt_device(y, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    H, fs_h, hop = efficient_hcqt_device(y, **kw)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"audio_seconds": seconds, "wall_s": dt, "audio_seconds_per_s": seconds / dt, "frames": int(H.shape[1]),
           "shape": list(H.shape), "parity": "unpinned (librosa absent; checked against oracle/restate_hcqt.py only)"}
    if with_cpu:
        from oracle import restate_hcqt as RH                       # checker timed as the CPU baseline, never shipped
        ys = y[: 2 * sr].cpu().numpy().astype(np.float64)
        t0 = time.perf_counter()
        RH.efficient_hcqt(ys, **kw)
        d2 = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": 2.0 / d2, "unit": "audio seconds/s", "cores": 1, "kind": "port",
                               "sample": "2 s of audio, numpy float64 restatement (direct filter bank, one core)"}
    return res


def segment_inference_probe(model, n_frames=2000, segment=100):
    """whole-recording inference, the reference's loop (one 75-frame patch per output frame,
    exp126a...py:427-443) next to the opt-in segment-wise path (windows of segment+74 frames, `segment` frames per
    forward).  Eval mode, the recording resident in HBM, predictions left on the device."""
    from multipitch_architectures_amd import experiment
    from multipitch_architectures_amd.synth import synth_file
    inputs, targets = synth_file(frames=n_frames, seed=21)
    was_training = model.training
    model.eval()
    res = {"frames": n_frames, "segment": segment}
    out = {}
    for key, seg in (("per_patch", None), ("segment_wise", segment)):
        experiment.predict_file(model, inputs, targets, segment=seg)         # warm-up: plans, packed filters
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out[key] = experiment.predict_file(model, inputs, targets, segment=seg)
        torch.cuda.synchronize()
        res[key + "_frames_per_s"] = n_frames / (time.perf_counter() - t0)
    d = (out["per_patch"] - out["segment_wise"]).abs()
    res["speedup"] = res["segment_wise_frames_per_s"] / res["per_patch_frames_per_s"]
    res["max_abs_diff"], res["mean_abs_diff"] = float(d.max()), float(d.mean())
    res["note"] = ("segment-wise is an approximation of the per-patch loop (zero padding at window instead of patch "
                   "borders); the weights are the randomly initialised ones after the timed steps, whose outputs are almost "
                   "constant (SURVEY section 4), so the deviation printed here is not informative -- "
                   "tests/test_gpu_experiment.py::test_segment_wise_inference exercises it with the deterministic fill")
    model.train(was_training)
    return res


def bf16x3_probe(model, criterion, opt, x, y, args, probe, kflops, gflop_table):
    """The same training step with the opt-in split-bf16 convolutions (ops.set_conv_precision("bf16x3")), timed after the
    headline region: a second, clearly labelled measurement -- never the line's `value`."""
    from multipitch_architectures_amd import ops
    from multipitch_architectures_amd.step import TrainStep
    ops.set_conv_precision("bf16x3")
    try:
        ts = TrainStep(model, criterion, opt, use_graph=not args.no_graph)
        for _ in range(3):
            ts(x, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ts(x, y)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        ops.set_kernel_probe(probe)
        for _ in range(PROBE_STEPS):
            ts.eager(x, y)
        pm = ops.probe_results_ms()
        ops.set_kernel_probe(None)
    finally:
        ops.set_conv_precision("f32")
    kms = sum(pm) / max(len(pm), 1)
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0
    ach = kflops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
    res = {"dtype": "bf16x3-f32acc", "ms_per_step": dt * 1e3, "value": args.global_batch * (args.frames - 74) / dt,
           "unit": "frames/s", "hip_graph": ts.graph is not None,
           "note": "opt-in (bench.py --conv-precision bf16x3): the 15-row convolutions as hi/lo bf16 halves, three bf16 "
                   "MFMAs per product, fp32 accumulation; everything else as in the headline run",
           "roofline": {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
                        "launch_ms": kms, "launches_timed": len(pm),
                        "kernel": "operand split + conv_bfx_kernel of the headline line's roofline layer",
                        "peak_note": "dense bf16 MFMA peak 2500 TFLOP/s / 3 MFMAs per product"}}
    if args.config in gflop_table and args.frames == 75:
        res["step_tflops"] = gflop_table[args.config] * args.global_batch / dt / 1e3
    return res


def t174_probe(model, criterion, opt, args, batch=64, frames=174):
    """SURVEY 8(d)'s secondary shape -- patches of 174 frames (100 output frames each; the shape of the north_star and of the
    reference's torchinfo summaries) -- as a labelled second measurement of the same model and step, timed after the
    headline region; never the line's `value`."""
    from multipitch_architectures_amd.configs import TRAIN_GFLOP_PER_PATCH_T174
    from multipitch_architectures_amd.step import TrainStep
    from multipitch_architectures_amd.synth import synth_batch
    dev = next(model.parameters()).device
    x, y = synth_batch(batch, frames, seed=1234)
    x, y = x.to(dev), y.to(dev)
    ts = TrainStep(model, criterion, opt, use_graph=not args.no_graph)
    for _ in range(3):
        ts(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    res = {"workload": f"{args.config} train step, batch {batch}, patches (6,{frames},216) -> ({frames - 74},72)",
           "global_batch": batch, "frames": frames, "ms_per_step": dt * 1e3, "patches_per_s": batch / dt,
           "value": batch * (frames - 74) / dt, "unit": "frames/s", "dtype": "f32", "hip_graph": ts.graph is not None}
    if args.config in TRAIN_GFLOP_PER_PATCH_T174 and frames == 174:
        res["step_tflops"] = TRAIN_GFLOP_PER_PATCH_T174[args.config] * batch / dt / 1e3
        res["step_mfma_frac"] = res["step_tflops"] / PEAK_FP32_MFMA_TFLOPS
    return res


def main():
    args = parse_args()
    # stdout carries exactly ONE line, the JSON: libraries that print banners to file descriptor 1 (RCCL's version block
    # at communicator creation) are sent to stderr for the whole run, the line goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    from multipitch_architectures_amd import nn_models, ops
    from multipitch_architectures_amd.configs import BASELINE_CONFIGS, CONFIGS, TRAIN_GFLOP_PER_PATCH
    from multipitch_architectures_amd.losses import BCELoss, PolyphonyLoss
    from multipitch_architectures_amd.optim import AdamW
    from multipitch_architectures_amd.parallel import GradientAverager, shard_range
    from multipitch_architectures_amd.step import TrainStep
    from multipitch_architectures_amd.synth import synth_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size must equal --gpus")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # one process per GPU; MPA_DIST_BACKEND=gloo + fewer devices than ranks is only for rehearsing the N>1 code path
    # on a single-GPU box (ranks then share a device and the collectives go through the host)
    backend = os.environ.get("MPA_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK={local_rank} but only {ndev} GPUs are visible")
    torch.cuda.set_device(local_rank % ndev)
    dev = torch.device("cuda", local_rank % ndev)
    dp = world > 1 or args.dp_rehearsal          # process group + gradient averager
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cfg = CONFIGS[args.config]
    baseline_batch = dict(BASELINE_CONFIGS)
    if args.global_batch is None:
        args.global_batch = baseline_batch.get(args.config, 256)
    torch.manual_seed(0)                                    # PyTorch default init, seed 0 (timing is value independent)
    model = getattr(nn_models, cfg["cls"])(**cfg["kwargs"]).to(dev).train()
    if dp:
        for p in model.parameters():
            dist.broadcast(p.data, 0)
    is_punet = cfg["cls"].endswith("polyphony_classif_softmax")
    loss_fn = PolyphonyLoss() if is_punet else BCELoss()
    opt = AdamW(model.parameters(), lr=cfg["lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    averager = GradientAverager(model.parameters()) if dp else None

    lo, hi = shard_range(args.global_batch, rank, world)
    x, y = synth_batch(args.global_batch, args.frames, seed=1234)
    x, y = x[lo:hi].to(dev), y[lo:hi].to(dev)               # inputs resident in HBM before the timed region
    ops.manual_seed(1234 + rank)
    ops.set_conv_precision(args.conv_precision)
    ops.set_data_parallel_exactness(sync_bn=args.sync_bn, gather_attention=args.gather_attention)
    bfx = args.conv_precision == "bf16x3"
    peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if bfx else PEAK_FP32_MFMA_TFLOPS

    criterion = (lambda res, t: loss_fn(res[0], res[1], t)) if is_punet else loss_fn
    # the reference's loop body; replayed as one captured HIP graph after the first (eager) steps unless --no-graph
    train_step = TrainStep(model, criterion, opt, averager=averager, use_graph=not args.no_graph)

    def step():
        return train_step(x, y)

    for _ in range(max(args.warmup, 2 if train_step.use_graph else 0)):   # eager step, then capture + first replay
        step()
    # Python's generation-2 collector walks every live object (modules, parameters, autograd graph) and costs ~70 ms
    # when it triggers -- two steps' worth at local batch 32.  Move what exists now to the permanent generation so that
    # collections inside the timed region only look at objects created by the steps themselves.
    gc.collect()
    gc.freeze()
    # live roofline probe: the dominant kernel = the forward convolution with the most algorithmic FLOPs (SAUnet:L:
    # upconv4.double_conv.4, 16->128 15x15 @75x216, 51 % of the model's MACs); input shapes recorded by forward hooks
    shapes = {}
    hooks = [m.register_forward_pre_hook(lambda mod, inp, nm=nm: shapes.__setitem__(nm, (mod, tuple(inp[0].shape))))
             for nm, m in model.named_modules() if isinstance(m, nn_models.layers.Conv2d)]
    model.eval()                       # (in training mode double_conv calls conv.forward_stats, which bypasses hooks)
    with torch.no_grad():
        model(x[:1])
    model.train()
    for h in hooks:
        h.remove()

    def conv_flops(mod, shp):
        oh = (shp[2] + 2 * mod.padding[0] - mod.kernel_size[0]) // mod.stride[0] + 1
        ow = (shp[3] + 2 * mod.padding[1] - mod.kernel_size[1]) // mod.stride[1] + 1
        return 2.0 * mod.out_channels * mod.in_channels * mod.kernel_size[0] * mod.kernel_size[1] * oh * ow, oh, ow
    dom_name, (dom, dshape) = max(shapes.items(), key=lambda kv: conv_flops(*kv[1])[0])
    dflops, dOH, dOW = conv_flops(dom, dshape)
    dkey = (dom.in_channels, dshape[2], dshape[3], dom.out_channels, dom.kernel_size[0], dom.kernel_size[1])
    probe = lambda k, kind: kind == "fwd" and tuple(k[1:7]) == dkey
    graphed = train_step.graph is not None
    if not graphed:          # kernel-by-kernel launches: HIP events bracket the dominant kernel inside the timed region
        ops.set_kernel_probe(probe)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        loss = step()
        host_s += time.perf_counter() - h0        # what the host spends enqueueing one step (no synchronisation inside)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    host_ms = [host_s / args.steps * 1e3]
    if world > 1:
        hm = [None] * world
        dist.all_gather_object(hm, host_ms[0])
        host_ms = hm
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if graphed:
        # a replayed graph cannot carry timing events around one of its kernels: the same launch (same tensors, same
        # plan) is timed right after the timed region in PROBE_STEPS kernel-by-kernel steps
        ops.set_kernel_probe(probe)
        for _ in range(PROBE_STEPS):
            train_step.eager(x, y)
    probe_ms = ops.probe_results_ms()
    ops.set_kernel_probe(None)
    # which device every rank actually ran on (gathered, so the line proves N distinct GPUs took part)
    props = torch.cuda.get_device_properties(dev)
    mine = {"rank": rank, "device": str(dev), "name": props.name, "pci_bus_id": getattr(props, "pci_bus_id", None)}
    devices = [mine]
    if world > 1:
        devices = [None] * world
        dist.all_gather_object(devices, mine)

    if rank == 0:
        ms = dt / args.steps * 1e3
        patches_per_s = args.global_batch / (dt / args.steps)
        frames_per_s = patches_per_s * (args.frames - 74)
        B_loc = hi - lo
        H, W = dshape[2], dshape[3]
        kflops = dflops * B_loc
        kms = sum(probe_ms) / max(len(probe_ms), 1)
        # HBM bytes per launch of that kernel and its MFMA-pipe busy fraction from separate rocprofv3 --pmc passes
        # (scratch/pmc_passes.sh -> scratch/pmc_summary.py -> profiles/); only valid for the configuration they were
        # collected on
        traffic = mfma_busy = traffic_source = None
        tfiles = ("r03_bf16x3_pmc_traffic.json",) if bfx else ("r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json")
        for tname in tfiles:
            tfile = os.path.join(ROOT, "profiles", tname)
            if os.path.exists(tfile) and args.config == "SAUnet:L" and B_loc == 256 and args.frames == 75:
                tj = json.load(open(tfile))
                traffic, mfma_busy = tj["traffic_bytes"], tj.get("mfma_busy")
                # not measured in this run: counters come from separate rocprofv3 --pmc passes (the pool forbids combining
                # them with anything else); the stamp says which file and which commit's kernels they describe
                traffic_source = {"file": "profiles/" + tname, "commit": tj.get("commit"), "kernel": tj.get("kernel"),
                                  "note": "separate rocprofv3 --pmc passes (scratch/pmc_passes.sh), not this run"}
                break
        achieved = kflops / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        out = {
            "metric": "HCQT frames/sec (train step), SAUnet:L" if args.config == "SAUnet:L" else f"HCQT frames/sec (train step), {args.config}",
            "value": frames_per_s, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "bf16x3-f32acc" if bfx else "f32",
            "data": "synthetic", "patches_per_s": patches_per_s, "loss": float(loss.detach()),
            "ranks_seen": dist.get_world_size() if dp else 1, "devices": devices,
            "dist_backend": backend if dp else None, "dp_rehearsal": bool(args.dp_rehearsal and world == 1),
            "config": {"workload": f"{args.config} ({cfg['cls']}) train step fwd+bwd+AdamW, global batch "
                                   f"{args.global_batch}, patches (6,{args.frames},216) -> ({args.frames - 74},72), "
                                   + (f"BASELINE.json configs[{[c for c, _ in BASELINE_CONFIGS].index(args.config)}]"
                                      if args.config in baseline_batch and args.global_batch == baseline_batch[args.config]
                                      else "not a BASELINE.json configuration"), "global_batch": args.global_batch,
                       "frames": args.frames, "parallelism": f"dp{world}"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "mfma_busy": mfma_busy,
                         "traffic_source": traffic_source,
                         "peak_note": ("dense bf16 MFMA peak 2500 TFLOP/s / 3 MFMAs per split-bf16 product; the launch "
                                       "includes the operand-split kernel in front of the convolution") if bfx else
                                      "dense fp32-input MFMA peak",
                         "kernel": f"{'conv_bfx_kernel' if bfx else 'conv_fwd_kernel'} {dom.in_channels}->{dom.out_channels} "
                                   f"{dom.kernel_size[0]}x{dom.kernel_size[1]} @{H}x{W} ({dom_name}), local batch {B_loc}",
                         "launch_ms": kms,
                         "launches_timed": len(probe_ms), "algorithmic_gflop_per_launch": kflops / 1e9,
                         "timed_in": "kernel-by-kernel steps right after the timed graph replays" if graphed
                                     else "the timed region"},
            "hip_graph": graphed,
            "host_enqueue_ms_per_step": host_ms,      # per rank: host time to enqueue one step (graph replays + collectives)
            "dp_graphs": bool(dp and graphed and train_step.graph_b is not None),
            # data-parallel replay: one graph per gradient bucket, bucket k's all-reduce launched behind segment k
            "dp_segments": [list(ids) for _, ids in train_step.segments] if dp and graphed else None,
            "dp_bucket_mb": [round(b["flat"].numel() * 4 / 2 ** 20, 2) for b in averager.buckets] if averager else None,
            "dp_exactness": {"sync_bn": bool(args.sync_bn), "gather_attention": bool(args.gather_attention)},
        }
        from multipitch_architectures_amd.configs import TRAIN_GFLOP_PER_PATCH_T174
        gflop_table = TRAIN_GFLOP_PER_PATCH if args.frames == 75 else (TRAIN_GFLOP_PER_PATCH_T174 if args.frames == 174 else {})
        if args.config in gflop_table:                                      # FLOP tables exist for T = 75 and T = 174 patches
            step_tflops = gflop_table[args.config] * patches_per_s / 1e3
            out["step_tflops"] = step_tflops
            out["step_mfma_frac"] = step_tflops / (PEAK_FP32_MFMA_TFLOPS * world)
            if bfx:
                out["step_mfma_frac_note"] = ("against the fp32-input MFMA peak (157.3): only the 15-row convolutions run on "
                                              "bf16 MFMA, so values above 1 are expected")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.config, args.frames)
        if world == 1 and not args.no_extras and not bfx and not args.dp_rehearsal and args.config == "SAUnet:L" \
                and args.frames == 75:
            out["t174"] = t174_probe(model, criterion, opt, args)
        if world == 1 and not args.no_extras and not bfx and not args.dp_rehearsal:
            out["bf16x3"] = bf16x3_probe(model, criterion, opt, x, y, args, probe, kflops, TRAIN_GFLOP_PER_PATCH)
        if world == 1 and not args.no_extras:
            out["patch_extraction"] = patch_extraction_probe(B_loc, not args.no_cpu_baseline)
            out["eval_measures"] = eval_measures_probe(not args.no_cpu_baseline)
            out["segment_inference"] = segment_inference_probe(model)
            out["hcqt_frontend"] = hcqt_probe(not args.no_cpu_baseline)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
