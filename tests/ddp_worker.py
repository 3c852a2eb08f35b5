"""One rank of the 2-rank data-parallel GPU tests (started by tests/test_gpu_parallel.py as a fresh process, RANK /
WORLD_SIZE / MASTER_* in the environment).  Prints one JSON line.

    python tests/ddp_worker.py <backend: nccl|gloo> <case: grads|step>

grads: tiny:CNN (no BatchNorm, no batch-axis attention; eval mode switches dropout off) -- the rank-averaged gradients
       of the two half batches must equal the gradients of the full batch computed by the same HIP model.
step : two data-parallel train steps of tiny:SAUnet (BN statistics and attention stay rank-local); parameters must stay
       bit-identical across the ranks.
graph: the same loop through step.TrainStep with an averager: forward + backward and the update replayed as two HIP graphs
       with the bucketed all-reduces between them; 5 steps, against the kernel-by-kernel data-parallel loop.
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
    elif case == "graph":
        from multipitch_architectures_amd.step import TrainStep
        runs = {}
        for use_graph in (False, True):
            model = build("tiny:SAUnet").train()
            ops.manual_seed(11 + rank)
            opt = AdamW(model.parameters(), lr=1e-3)
            avg = GradientAverager(model.parameters(), bucket_bytes=1 << 16)
            ts = TrainStep(model, BCELoss(), opt, averager=avg, use_graph=use_graph)
            losses = [float(ts(x[lo:hi], y[lo:hi])) for _ in range(5)]
            flat = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
            both = [torch.empty_like(flat) for _ in range(world)]
            dist.all_gather(both, flat)
            runs[use_graph] = dict(losses=losses, flat=flat, identical=all(bool(torch.equal(both[0], b)) for b in both[1:]),
                                   replays=ts.replays, graphs=(ts.graph is not None, ts.graph_b is not None))
            avg.remove()
        a, b = runs[False]["flat"], runs[True]["flat"]
        out.update(losses=runs[True]["losses"], losses_eager=runs[False]["losses"],
                   params_identical=runs[True]["identical"] and runs[False]["identical"],
                   finite=bool(torch.isfinite(b).all()), replays=runs[True]["replays"], graphs=runs[True]["graphs"],
                   graph_vs_eager=float((a - b).abs().max() / a.abs().max()), opt_steps=int(next(iter(opt.state.values()))["step"]))
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
