"""One training step -- forward, loss, backward, (gradient averaging,) AdamW -- as a replayable HIP graph.

The reference's loop body (exp126a_musicnet_cnn_basic.py:318-327)

    y_pred = model(local_batch); loss = criterion(y_pred, local_labels)
    optimizer.zero_grad(); loss.backward(); optimizer.step()

is ~520 kernel launches for SAUnet:L.  At the local batch of an 8-GPU run (32 patches) the kernels take ~30 ms and the
launch-to-launch gaps another ~1.3 ms; replaying the captured step removes the host from the loop entirely.  Everything
that varies between steps is device-resident (dropout stream position, AdamW step count and learning rate, BatchNorm
running statistics), so one capture serves the whole run.

    step = TrainStep(model, criterion, optimizer)        # criterion(model_output, y) -> scalar loss
    for x, y in loader:
        loss = step(x, y)                                 # static tensor; float(loss) syncs, as loss.item() does

Batches whose shape differs from the captured one (the last, smaller batch of an epoch) run eagerly.

Data-parallel ranks (an ``averager``) replay the step as a *chain of graphs with the collectives between them*, so
that the RCCL all-reduces overlap the backward pass although no collective sits inside a captured graph:

    segment 0 = forward + loss + backward down to the last gradient of bucket 0 + that bucket's gather
    -> all-reduce of bucket 0 goes on RCCL's stream (it waits for segment 0 only)
    segment 1 = backward down to the last gradient of bucket 1 + gather      (runs while bucket 0 is on the wire)
    -> all-reduce of bucket 1 ...
    last segment = the rest of backward (+ buckets no gradient reported to) -> the last, small bucket (parallel.py)
    update graph = 1/world scale + AdamW + dropout-stream advance            (after the compute stream has waited for
                                                                              the collectives)

The cut points are found while capturing: the hook of a bucket's last gradient (`GradientAverager.on_bucket`, on
autograd's device thread) ends the running capture and begins the next one on the same stream and memory pool -- which
needs ``capture_error_mode="relaxed"`` (a stream capture may otherwise only be ended by the thread that began it).  A
rank issues (number of buckets + 1) graph launches and as many collectives per step instead of ~480 kernel launches:
the host stays off the critical path when eight ranks share one CPU.
"""
import gc
import os

import torch

from . import ops


class TrainStep:
    def __init__(self, model, criterion, optimizer, averager=None, use_graph=True):
        self.model, self.criterion, self.opt, self.averager = model, criterion, optimizer, averager
        # collectives stay outside captured graphs (see parallel.py): with an averager the step is two graphs
        self.use_graph = bool(use_graph)
        self.graph = None
        self.graph_b = None          # data-parallel: scale + optimizer update
        self.segments = []           # data-parallel: [(graph, bucket indices to put on the wire after it)]; graph = segment 0
        self.shape = None            # shapes the graph was captured for
        self._eager_shape = None     # shapes of the last eager step (the capture follows an eager step of its shape)
        self._x = self._y = self._loss = None
        self.replays = 0
        self._pack_tables = {}       # (x shape, y shape) -> ops.PackTable of the filter banks that shape's step uses
        self._old_tables = []        # tables a live graph may still launch on (cleared when the graph is dropped)
        self._opt_epoch = None       # optimizer.table_epoch the graph was captured under
        self._no_graph_shapes = set()   # shapes whose capture failed: they keep running kernel by kernel
        self._use_pack_tables = os.environ.get("MPA_PACK_TABLES", "1") != "0"      # diagnostics: "0" = pack bank by bank

    # the loop body in three phases (launched kernel by kernel, or captured: A = _fwd_bwd, B = _update)
    def _fwd_bwd(self, x, y):
        # every filter bank this shape's step uses, re-packed for the current weights in one launch (the table is known
        # from the previous step of the shape; the first one packs lazily, bank by bank)
        shape = (tuple(x.shape), tuple(y.shape))
        tab = self._pack_tables.get(shape) if self._use_pack_tables else None
        if tab is not None and not ops.pack_table_valid(tab):
            # a weight was re-allocated (model.to(), load_state_dict(assign=True), p.data = ...): the table's baked-in
            # addresses are stale -- pack lazily this step and build a new table at its end
            self._pack_tables.pop(shape)
            tab = None
        if tab is not None:
            ops.run_pack_table(tab)
        ops.begin_pack_window()
        loss = self.criterion(self.model(x), y)
        self.opt.zero_grad()
        loss.backward(ops.backward_seed(loss))
        return loss

    def _update(self):
        self.opt.step()
        ops.rng_advance()

    def _bookkeep(self, shape):
        tab = self._pack_tables.get(shape) if self._use_pack_tables else None
        keys = ops.pack_window_keys() if self._use_pack_tables else None
        if self._use_pack_tables and (tab is None or tab.keys != keys):
            if tab is not None:
                self._old_tables.append(tab)         # a captured graph may still launch on it
            self._pack_tables[shape] = ops.build_pack_table(keys)

    def eager(self, x, y):
        capturing = torch.cuda.is_current_stream_capturing()
        loss = self._fwd_bwd(x, y)
        if self.averager is not None:
            self.averager.finish()
        self._update()
        if not capturing:
            self._bookkeep((tuple(x.shape), tuple(y.shape)))
        # detached: a caller holding the loss across iterations (every training loop does) must not keep this step's
        # autograd nodes alive -- stale AccumulateGrad nodes would run on the stream they were created on and break the
        # capture of the next step
        return loss.detach()

    def _capture(self, x, y):
        # No throw-away warm-up steps (they would be extra optimiser steps the reference's loop does not take): the
        # capture is taken on the second step of a shape, after an ordinary eager step has created everything that is
        # set up lazily -- optimizer state, the dropout stream state, kernel attributes.
        self._x, self._y = x.clone(), y.clone()
        self.opt.sync_hyper()
        self.opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        rng_local = ops._Rng.local
        av = self.averager
        try:
            if av is None:
                with torch.cuda.graph(graph):
                    self._loss = self.eager(self._x, self._y)
            else:
                graph = self._capture_segments(self._x, self._y)
        except Exception:
            # nothing of the failed capture ran on the device; restore the host-side bookkeeping it touched and let the
            # caller run this shape kernel by kernel from now on
            ops._Rng.local = rng_local
            self._x = self._y = self._loss = None
            self.graph = self.shape = self.graph_b = None
            self.segments = []
            if av is not None:
                for b in av.buckets:
                    b["pending"], b["gathered"], b["handle"] = len(b["params"]), False, None
            self.opt.zero_grad(set_to_none=True)
            self.opt.invalidate_grad_table()
            raise
        self.graph, self.shape = graph, (tuple(x.shape), tuple(y.shape))
        self._opt_epoch = getattr(self.opt, "table_epoch", None)

    def _capture_segments(self, x, y):
        """data-parallel capture: forward + backward as one graph per gradient bucket (module docstring), then the update
        graph.  Returns segment 0."""
        av = self.averager
        torch.cuda.synchronize()
        gc.collect()
        torch.cuda.empty_cache()                      # (what torch.cuda.graph() does on entry)
        stream = torch.cuda.Stream()
        stream.wait_stream(torch.cuda.current_stream())
        pool = torch.cuda.graph_pool_handle()
        segments = []                                 # [graph, [bucket indices]]
        waiting = {"n": len(av.buckets)}

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=pool, capture_error_mode="relaxed")
            segments.append([g, []])

        def on_bucket(index):                         # autograd's device thread, inside loss.backward()
            segments[-1][1].append(index)
            waiting["n"] -= 1
            if waiting["n"] > 0:                      # (the last bucket's segment also takes the rest of backward)
                segments[-1][0].capture_end()
                begin()

        av.deferred, av.on_bucket = True, on_bucket
        try:
            with torch.cuda.stream(stream):
                begin()
                try:
                    self._loss = self._fwd_bwd(x, y).detach()
                    av.gather_remaining()             # buckets some parameter never reported to
                    segments[-1][0].capture_end()
                except BaseException:
                    try:                              # never leave the stream capturing
                        segments[-1][0].capture_end()
                    except Exception:                 # noqa: BLE001
                        pass
                    raise
                seen = {i for _, ids in segments for i in ids}
                segments[-1][1] += [b["index"] for b in av.buckets if b["index"] not in seen]
                # a capture records, it does not run: nothing has been computed yet.  The update graph is captured right
                # away (its inputs -- the flat buffers the gradients are averaged in -- have fixed addresses).
                av.expose()
                graph_b = torch.cuda.CUDAGraph()
                graph_b.capture_begin(pool=pool, capture_error_mode="relaxed")
                try:
                    av.scale_all()
                    self._update()
                finally:
                    graph_b.capture_end()
        finally:
            av.deferred, av.on_bucket = False, None
            torch.cuda.current_stream().wait_stream(stream)
        self.segments = [(g, tuple(ids)) for g, ids in segments]
        self.graph_b = graph_b
        return segments[0][0]

    def _agree(self, ok, device):
        """data-parallel ranks replay graphs only if every rank captured: a rank that fell back to the kernel-by-kernel
        loop on its own would still pair its collectives with the others' (same buckets, same order), but the decision
        is made once, together"""
        av = self.averager
        if av is None or av.world <= 1:
            return ok
        import torch.distributed as dist
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=av.group)
        return bool(flag.item())

    def _drop_graph(self):
        self.graph = self.shape = self.graph_b = None
        self.segments = []
        self._x = self._y = self._loss = None
        self._eager_shape = None
        self._old_tables.clear()

    def __call__(self, x, y):
        if not self.use_graph or not self.model.training or ops.exactness_collectives_active():
            # (SyncBN / gathered attention put collectives inside forward and backward: never captured)
            # the captured graph is the *training* step (dropout, batch statistics, running-stat updates): after
            # model.eval() the loop body runs kernel by kernel with whatever mode the modules are in
            return self.eager(x, y)
        shape = (tuple(x.shape), tuple(y.shape))
        if self.graph is not None:
            tab = self._pack_tables.get(self.shape)
            if self._opt_epoch != getattr(self.opt, "table_epoch", None) or \
                    (tab is not None and not ops.pack_table_valid(tab)):
                # optimizer.load_state_dict() / add_param_group(), or re-allocated weights: the graph has the old
                # exp_avg / step-count / weight addresses baked in -- capture again after one eager step
                self._drop_graph()
        if self.graph is None:
            if shape != self._eager_shape or shape in self._no_graph_shapes:
                self._eager_shape = shape
                return self.eager(x, y)
            try:
                self._capture(x, y)
                ok = True
            except RuntimeError:
                ok = False
            if not self._agree(ok, x.device):
                if ok:
                    self._drop_graph()
                    self.opt.zero_grad(set_to_none=True)
                    self.opt.invalidate_grad_table()
                self._no_graph_shapes.add(shape)
                return self.eager(x, y)
        if shape != self.shape:
            return self.eager(x, y)
        if x.data_ptr() != self._x.data_ptr():
            self._x.copy_(x)
        if y.data_ptr() != self._y.data_ptr():
            self._y.copy_(y)
        self.opt.sync_hyper()                        # ReduceLROnPlateau may have changed the learning rate
        if self.averager is None:
            self.graph.replay()
        else:
            # one graph per gradient bucket, each bucket on the wire as soon as its segment is enqueued: the collective
            # waits (on RCCL's stream) for that segment only and runs under the following ones
            av = self.averager
            for g, bucket_ids in self.segments:
                g.replay()
                for i in bucket_ids:
                    av.launch(i)
            av.wait_all()                            # the compute stream waits; the host does not (RCCL)
            self.graph_b.replay()
        self.replays += 1
        self.opt.note_steps(1)                       # host mirrors of what the graph did on the device
        self.opt.invalidate_grad_table()             # the replay left the graph's gradient addresses in the device table
        ops.bump_param_epoch()
        return self._loss

    def static_inputs(self):
        """the graph's input tensors: fill them in place to skip the copy in __call__"""
        return self._x, self._y
