"""One rank of the 2-rank data-parallel GPU tests (started by tests/test_gpu_parallel.py as a fresh process, RANK /
WORLD_SIZE / MASTER_* in the environment).  Prints one JSON line.

    python tests/ddp_worker.py <backend: nccl|gloo> <case: grads|step>

grads: tiny:CNN (no BatchNorm, no batch-axis attention; eval mode switches dropout off) -- the rank-averaged gradients
       of the two half batches must equal the gradients of the full batch computed by the same HIP model.
step : two data-parallel train steps of tiny:SAUnet (BN statistics and attention stay rank-local); parameters must stay
       bit-identical across the ranks.
graph: the same loop through step.TrainStep with an averager: forward + backward replayed as one HIP graph per gradient
       bucket with each bucket's all-reduce launched behind its segment (overlapping the following ones), then the update
       graph; 5 steps, against the kernel-by-kernel data-parallel loop.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    backend, case = sys.argv[1], sys.argv[2]
    from multipitch_architectures_amd import nn_models, ops
    from multipitch_architectures_amd.configs import CONFIGS
    from multipitch_architectures_amd.losses import BCELoss
    from multipitch_architectures_amd.optim import AdamW
    from multipitch_architectures_amd.parallel import GradientAverager, shard_range
    from multipitch_architectures_amd.synth import det_fill, synth_batch

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]) % ndev)
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)

    def build(name):
        cfg = CONFIGS[name]
        m = getattr(nn_models, cfg["cls"])(**cfg["kwargs"])
        m.load_state_dict(det_fill(m.state_dict()))
        return m.to(dev)

    B = 8
    x, y = synth_batch(B, 75, seed=5)
    x, y = x.to(dev), y.to(dev)
    lo, hi = shard_range(B, rank, world)
    out = {"rank": rank, "world": dist.get_world_size(), "device": str(dev)}
    if case == "grads":
        model = build("tiny:CNN").eval()
        BCELoss()(model(x), y).backward()
        full = [p.grad.clone() for p in model.parameters()]
        model.zero_grad(set_to_none=True)
        avg = GradientAverager(model.parameters(), bucket_bytes=2048)
        errs = []
        for _ in range(2):                         # twice: the buckets re-arm
            model.zero_grad(set_to_none=True)
            BCELoss()(model(x[lo:hi]), y[lo:hi]).backward()
            avg.finish()
            errs.append(max(float((p.grad - g).abs().max() / g.abs().max().clamp_min(1e-30))
                            for p, g in zip(model.parameters(), full)))
        out.update(buckets=len(avg.buckets), rel_err=max(errs))
    elif case == "exact":
        # SURVEY 8e exactness modes: with SyncBN and gathered keys / values a 2 x B/2 run reproduces the 1 x B run
        from multipitch_architectures_amd.nn_models.layers import Dropout
        model = build("tiny:SAUnet").train()
        for m in model.modules():
            if isinstance(m, Dropout):
                m.p = 0.0
            if hasattr(m, "p_dropout"):
                m.p_dropout = 0.0
        sd0 = {k: v.clone() for k, v in model.state_dict().items()}
        full_loss = BCELoss()(model(x), y)
        full_loss.backward()
        full = [p.grad.clone() for p in model.parameters()]
        model.load_state_dict(sd0)                       # (running statistics moved: restore)
        model.zero_grad(set_to_none=True)
        res = {}
        for mode in ("local", "exact"):
            ops.set_data_parallel_exactness(sync_bn=mode == "exact", gather_attention=mode == "exact")
            model.load_state_dict(sd0)
            model.zero_grad(set_to_none=True)
            avg = GradientAverager(model.parameters(), bucket_bytes=1 << 16)
            loss = BCELoss()(model(x[lo:hi]), y[lo:hi])
            loss.backward()
            avg.finish()
            lsum = loss.detach().clone()
            dist.all_reduce(lsum)
            errs = [float((p.grad - g).abs().max() / g.abs().max().clamp_min(1e-30)) for p, g in zip(model.parameters(), full)
                    if float(g.abs().max()) > 1e-6]
            res[mode] = dict(loss_err=abs(float(lsum) / world - float(full_loss)), grad_err=max(errs),
                             grad_err_median=sorted(errs)[len(errs) // 2])
            avg.remove()
        ops.set_data_parallel_exactness()
        rm = [v for k, v in model.state_dict().items() if k.endswith("running_mean")][0]
        out.update(res=res, full_loss=float(full_loss))
    elif case == "graph":
        from multipitch_architectures_amd.step import TrainStep
        def run(name, use_graph):
            model = build(name).train()
            ops.manual_seed(11 + rank)
            opt = AdamW(model.parameters(), lr=1e-3)
            avg = GradientAverager(model.parameters(), bucket_bytes=1 << 16)
            ts = TrainStep(model, BCELoss(), opt, averager=avg, use_graph=use_graph)
            losses = [float(ts(x[lo:hi], y[lo:hi])) for _ in range(5)]
            flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
            both = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            avg.remove()
            return dict(losses=losses, flat=flat, identical=all(bool(torch.equal(both[0], b)) for b in both[1:]),
                        replays=ts.replays, graphs=(ts.graph is not None, ts.graph_b is not None),
                        segments=[list(ids) for _, ids in ts.segments], buckets=len(avg.buckets),
                        opt_steps=int(next(iter(opt.state.values()))["step"]))
        runs = {g: run("tiny:SAUnet", g) for g in (False, True)}
        cnn = {g: run("tiny:CNN", g) for g in (False, True)}
        a, b = runs[False]["flat"], runs[True]["flat"]
        ca, cb = cnn[False]["flat"], cnn[True]["flat"]
        out.update(losses=runs[True]["losses"], losses_eager=runs[False]["losses"],
                   params_identical=runs[True]["identical"] and runs[False]["identical"] and cnn[True]["identical"],
                   finite=bool(torch.isfinite(b).all()), replays=runs[True]["replays"], graphs=runs[True]["graphs"],
                   segments=runs[True]["segments"], buckets=runs[True]["buckets"],
                   graph_vs_eager=float((a - b).abs().max() / a.abs().max()), opt_steps=runs[True]["opt_steps"],
                   cnn_graph_vs_eager=float((ca - cb).abs().max() / ca.abs().max()),
                   cnn_loss_dev=max(abs(u - v) for u, v in zip(cnn[True]["losses"], cnn[False]["losses"])))
    else:
        model = build("tiny:SAUnet").train()
        ops.manual_seed(11 + rank)
        opt = AdamW(model.parameters(), lr=1e-3)
        avg = GradientAverager(model.parameters(), bucket_bytes=1 << 16)
        losses = []
        for _ in range(2):
            loss = BCELoss()(model(x[lo:hi]), y[lo:hi])
            opt.zero_grad()
            loss.backward()
            avg.finish()
            opt.step()
            losses.append(float(loss))
        flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        out.update(losses=losses, params_identical=all(bool(torch.equal(both[0], b)) for b in both[1:]),
                   finite=bool(torch.isfinite(flat).all()))
    torch.cuda.synchronize()
    print("DDP_RESULT " + json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
