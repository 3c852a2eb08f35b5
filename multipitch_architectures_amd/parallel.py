"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL all-reduce of gradients over xGMI,
bucketed and overlapped with the rest of backward.

The reference is single-GPU (exp126a_musicnet_cnn_basic.py:209-212: ``device = cuda:0``); this is build-side
functionality required by BASELINE.json.  Semantics: the batch is sharded over ranks,
BatchNorm statistics and the batch-axis attention stay *local* to a rank (exactly what wrapping the reference in
``DistributedDataParallel`` would do), gradients are averaged over ranks.

Why buckets sized like this: xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce of G bytes
moves 2(N-1)/N * G per GPU over one link pair, i.e. ~0.4 ms for SAUnet:L's 32.5 MB -- against ~18 ms of backward at the
local batch of an 8-GPU run.  A handful of large buckets (default 8 MB) keeps every collective bandwidth-bound rather
than latency-bound, and since the head's and the decoder's gradients are produced first, their buckets are on the wire
while the encoder is still back-propagating.  Only the last bucket cannot overlap anything, so it is kept small
(`tail_bytes`, default 2 MB: the first encoder layers -- 1.7 MB for SAUnet:L).

Two ways a step drives this class:
* kernel by kernel (`finish()` after `loss.backward()`): the hooks put each bucket on the wire the moment its last
  gradient has been accumulated;
* `step.TrainStep` replaying the step as HIP graphs: the backward pass is captured as one graph *segment per bucket*
  (the hook of a bucket's last gradient cuts the capture, `on_bucket`), and the replay launches bucket k's all-reduce
  between segment k and segment k+1 -- same overlap, no collective inside a captured graph.
"""
import torch
import torch.distributed as dist

from . import _lib as L


def shard_range(global_batch: int, rank: int, world: int):
    """rank r owns patches [r*B/W, (r+1)*B/W); B must divide evenly (strong scaling of a fixed global batch)."""
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


class GradientAverager:
    """Bucketed, overlapped gradient all-reduce.  Usage:

        avg = GradientAverager(model.parameters())
        loss.backward()            # hooks launch one async all-reduce per bucket as soon as it is complete
        avg.finish()               # wait, scale by 1/world, expose the averaged values as p.grad
        optimizer.step()
    """

    def __init__(self, params, bucket_bytes: int = 8 << 20, process_group=None, tail_bytes: int = 2 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.buckets = []        # list of dict(params, flat, views, pending, handle)
        self._owner = {}
        # deferred: the hooks only gather a completed bucket into its flat buffer and leave the collective to
        # launch_all() -- what `step.TrainStep` sets while it captures forward + backward as a HIP graph (no collective
        # inside a captured graph; they run between the step's two graphs)
        self.deferred = False
        # deferred mode: called with the bucket's index right after its gather (on autograd's device thread) -- where
        # `step.TrainStep` cuts its capture
        self.on_bucket = None
        self.bucket_bytes, self.tail_bytes = int(bucket_bytes), int(tail_bytes)
        # buckets in reverse registration order first (~ the order in which autograd produces gradients); the first
        # backward pass records the real arrival order and finish() re-buckets by it once (every rank sees the same
        # order: same model, same autograd graph), so that bucket k really completes before bucket k+1
        self._arrival, self._rebucketed = [], False
        self._build_buckets(list(reversed(self.params)))
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in self.params]

    def _build_buckets(self, ordered):
        bucket_bytes, tail_bytes = self.bucket_bytes, self.tail_bytes
        self.buckets, self._owner = [], {}
        plists, cur, cur_bytes = [], [], 0
        for p in ordered:
            nbytes = p.numel() * p.element_size()
            if cur and cur_bytes + nbytes > bucket_bytes:
                plists.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            plists.append(cur)
        # the last bucket is the one whose all-reduce nothing can hide: cap it at tail_bytes (the parameters registered
        # first, i.e. whose gradients arrive last)
        if plists and tail_bytes and tail_bytes < bucket_bytes:
            last, tail, nb = plists[-1], [], 0
            while len(last) > 1 and nb + last[-1].numel() * last[-1].element_size() <= tail_bytes:
                nb += last[-1].numel() * last[-1].element_size()
                tail.insert(0, last.pop())
            if tail:
                plists.append(tail)
        for pl in plists:
            self._add_bucket(pl)

    def _add_bucket(self, plist):
        n = sum(p.numel() for p in plist)
        flat = torch.zeros(n, dtype=plist[0].dtype, device=plist[0].device)
        views, off = [], 0
        for p in plist:
            views.append(flat[off:off + p.numel()].view_as(p))
            off += p.numel()
        offs, off = [], 0
        for p in plist:
            offs.append(off)
            off += p.numel()
        b = {"params": plist, "flat": flat, "views": views, "pending": len(plist), "handle": None, "offsets": offs,
             "gathered": False, "index": len(self.buckets)}
        for p in plist:
            self._owner[p] = b
        self.buckets.append(b)

    def _on_grad(self, p):
        if not self._rebucketed and len(self._arrival) < len(self.params):     # (the first backward pass only)
            self._arrival.append(p)
        b = self._owner[p]
        b["pending"] -= 1
        if b["pending"] == 0:
            # the bucket is complete: gather its gradients into the flat buffer (one launch per 32 tensors instead of one
            # copy per parameter -- 168 of them for SAUnet:L) and put it on the wire
            self._gather(b)
            b["gathered"] = True
            if self.deferred:
                if self.on_bucket is not None:
                    self.on_bucket(b["index"])
            elif self.world > 1:
                b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    @staticmethod
    def _gather(b):
        plist = [(i, p) for i, p in enumerate(b["params"]) if p.grad is not None]
        if not plist:
            return
        if not b["flat"].is_cuda:           # gloo / CPU rehearsal of the communication logic only
            for i, p in plist:
                b["views"][i].copy_(p.grad)
            return
        import ctypes
        n = len(plist)
        for _, p in plist:
            if not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise RuntimeError("GradientAverager needs contiguous fp32 gradients")
        srcs = (ctypes.c_void_p * n)(*[p.grad.data_ptr() for _, p in plist])
        offs = (ctypes.c_int64 * n)(*[b["offsets"][i] for i, _ in plist])
        sizes = (ctypes.c_int64 * n)(*[p.numel() for _, p in plist])
        rc = L.load().mpa_gather_copy(ctypes.c_void_p(b["flat"].data_ptr()), srcs, offs, sizes, n,
                                      ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        L.check(rc, "mpa_gather_copy")

    # finish() = gather_remaining() + launch_all() + wait_all() + scale_all() + expose(); the pieces are what the
    # graph-captured step calls separately (gather inside the first graph, collectives between the graphs, scale inside
    # the second one)
    def gather_remaining(self):
        """buckets that some parameter never reported to (no gradient this step): zeros for those, then gather"""
        for b in self.buckets:
            if b["pending"] != 0 and not b.get("gathered"):
                for i, p in enumerate(b["params"]):
                    if p.grad is None:
                        b["views"][i].zero_()
                self._gather(b)
                b["gathered"] = True

    def launch(self, index):
        """bucket `index` on the wire (asynchronous: the collective waits for what the current stream holds so far)"""
        b = self.buckets[index]
        if self.world > 1 and b["handle"] is None:
            b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def launch_all(self):
        """one asynchronous all-reduce per bucket that is not on the wire yet"""
        if self.world > 1:
            for b in self.buckets:
                if b["handle"] is None:
                    b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait_all(self):
        for b in self.buckets:
            if b["handle"] is not None:
                b["handle"].wait()
                b["handle"] = None

    def scale_all(self):
        if self.world > 1:
            for b in self.buckets:
                self._scale(b["flat"], 1.0 / self.world)

    def expose(self):
        """the averaged values become p.grad (views of the flat buffers); re-arm the buckets for the next step"""
        for b in self.buckets:
            for p, v in zip(b["params"], b["views"]):
                p.grad = v
            b["pending"] = len(b["params"])
            b["gathered"] = False

    def finish(self):
        self.gather_remaining()
        for b in self.buckets:
            if b["handle"] is None and self.world > 1 and not self.deferred:
                b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.wait_all()
        self.scale_all()
        self.expose()
        if not self._rebucketed:
            self._rebucket()

    def _rebucket(self):
        """after the first backward pass: buckets in the order the gradients really arrived (parameters that got no
        gradient go last).  The gradients just exposed are views of the old flat buffers and stay valid; the next step
        gathers into the new ones."""
        self._rebucketed = True
        seen, order = set(), []
        for p in self._arrival:
            if id(p) not in seen:
                seen.add(id(p))
                order.append(p)
        order += [p for p in reversed(self.params) if id(p) not in seen]
        self._arrival = []
        if [id(p) for b in self.buckets for p in b["params"]] != [id(p) for p in order]:
            self._build_buckets(order)

    @staticmethod
    def _scale(flat, alpha):
        if flat.is_cuda:
            import ctypes
            rc = L.load().mpa_scale(alpha, ctypes.c_void_p(flat.data_ptr()), flat.numel(),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            L.check(rc, "mpa_scale")
        else:                     # gloo / CPU rehearsal of the communication logic only
            flat.mul_(alpha)

    def remove(self):
        for h in self._hooks:
            h.remove()
