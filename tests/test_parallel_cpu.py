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


class _LateFirst(torch.nn.Module):
    """registered order != gradient order: `tail` is registered first but used last in forward (so its gradient arrives
    first), `head` is registered last and used first (gradient last)"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.tail = torch.nn.Linear(40, 3)
        self.mid = torch.nn.Linear(40, 40)
        self.head = torch.nn.Linear(12, 40)

    def forward(self, x):
        return self.tail(torch.tanh(self.mid(torch.tanh(self.head(x)))))


def _worker_segments(rank, world, port, q):
    """what step.TrainStep does with the averager while it captures: deferred hooks report every completed bucket through
    on_bucket (the capture is cut there) and the replay launches bucket k's all-reduce behind segment k"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _LateFirst()
    torch.manual_seed(1)
    x, y = torch.randn(8, 12), torch.randn(8, 3)
    lo, hi = shard_range(8, rank, world)
    avg = GradientAverager(model.parameters(), bucket_bytes=4096, tail_bytes=1024)
    first_layout = [[id(p) for p in b["params"]] for b in avg.buckets]
    # step 1, kernel by kernel: records the arrival order, then re-buckets by it
    ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
    avg.finish()
    g1 = [p.grad.clone() for p in model.parameters()]
    arrival = [id(p) for b in avg.buckets for p in b["params"]]
    tail_bytes = sum(p.numel() * 4 for p in avg.buckets[-1]["params"])
    # step 2, the way the captured step drives it
    order = []
    model.zero_grad(set_to_none=True)
    avg.deferred, avg.on_bucket = True, order.append
    ((model(x[lo:hi]) - y[lo:hi]) ** 2).mean().backward()
    avg.gather_remaining()
    avg.deferred, avg.on_bucket = False, None
    for i in order:                       # bucket k on the wire "behind segment k"
        avg.launch(i)
    avg.wait_all()
    avg.scale_all()
    avg.expose()
    same = all(torch.allclose(a, p.grad, atol=1e-7) for a, p in zip(g1, model.parameters()))
    ids = {id(p): n for n, p in model.named_parameters()}
    q.put((rank, dict(order=order, n_buckets=len(avg.buckets), same=same, tail_bytes=tail_bytes,
                      rebucketed=first_layout != [[id(p) for p in b["params"]] for b in avg.buckets],
                      first_bucket=[ids[i] for i in arrival[:2]], last_bucket=[ids[id(p)] for p in avg.buckets[-1]["params"]])))
    dist.destroy_process_group()


def test_buckets_follow_the_arrival_order_and_report_in_it():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_segments, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        g = got[r]
        assert g["rebucketed"] and g["same"], g                   # the registration order was wrong for this model
        assert g["order"] == list(range(g["n_buckets"])), g       # after re-bucketing, bucket k completes k-th
        assert set(g["first_bucket"]) == {"tail.weight", "tail.bias"}, g      # the gradients that arrive first
        assert g["last_bucket"][-1].startswith("head.") and g["tail_bytes"] <= 4096, g     # ... and last
    assert got[0]["order"] == got[1]["order"]
