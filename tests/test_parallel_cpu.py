"""world_size-2 gloo rehearsal of the data-parallel path on CPU: sharding, bucketing, hook-driven async all-reduce,
averaging, and that the averaged gradients equal the single-process gradients of the full batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import multiprocessing as mp

from multipitch_architectures_amd.parallel import GradientAverager, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 40), torch.nn.Tanh(),
                               torch.nn.Linear(40, 3))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _toy()
    torch.manual_seed(1)
    x, y = torch.randn(8, 12), torch.randn(8, 3)
    lo, hi = shard_range(8, rank, world)
    avg = GradientAverager(model.parameters(), bucket_bytes=4096)        # several buckets
    assert len(avg.buckets) > 1
    for _ in range(2):                                                     # two steps: buckets are re-armed
        model.zero_grad(set_to_none=True)
        # mean over the *global* batch = average over ranks of the local means (equal shard sizes)
        loss = ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean()
        loss.backward()
        avg.finish()
    q.put((rank, [p.grad.numpy().copy() for p in model.parameters()]))     # by value: no fd passing
    dist.destroy_process_group()


def test_gradient_averager_matches_full_batch():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _toy()
    torch.manual_seed(1)
    x, y = torch.randn(8, 12), torch.randn(8, 3)
    ((model(x) - y) ** 2).mean().backward()
    for r in range(world):
        for g, p in zip(got[r], model.parameters()):
            assert torch.allclose(torch.from_numpy(g), p.grad, atol=1e-6), r
    for a, b in zip(got[0], got[1]):
        assert (a == b).all()


def _worker_deferred(rank, world, port, q):
    """the split interface the graph-captured step uses: hooks gather only (deferred), collectives launched afterwards"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _toy()
    torch.manual_seed(1)
    x, y = torch.randn(8, 12), torch.randn(8, 3)
    lo, hi = shard_range(8, rank, world)
    avg = GradientAverager(model.parameters(), bucket_bytes=4096)
    for _ in range(2):
        model.zero_grad(set_to_none=True)
        avg.deferred = True
        ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
        avg.gather_remaining()
        avg.deferred = False
        assert all(b["handle"] is None for b in avg.buckets)            # nothing went on the wire during backward
        avg.launch_all()
        avg.wait_all()
        avg.scale_all()
        avg.expose()
    q.put((rank, [p.grad.numpy().copy() for p in model.parameters()]))
    dist.destroy_process_group()


def test_deferred_collectives_give_the_same_average():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_deferred, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _toy()
    torch.manual_seed(1)
    x, y = torch.randn(8, 12), torch.randn(8, 3)
    ((model(x) - y) ** 2).mean().backward()
    for r in range(world):
        for g, p in zip(got[r], model.parameters()):
            assert torch.allclose(torch.from_numpy(g), p.grad, atol=1e-6), r


def test_shard_range():
    assert [shard_range(256, r, 8) for r in (0, 7)] == [(0, 32), (224, 256)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_single_process_averager_is_a_no_op():
    model = _toy()
    avg = GradientAverager(model.parameters())
    x = torch.randn(4, 12)
    model(x).sum().backward()
    ref = [p.grad.clone() for p in model.parameters()]
    avg.finish()
    for g, p in zip(ref, model.parameters()):
        assert torch.equal(g, p.grad)
