"""torch.autograd.Function wrappers over the C ABI of libmpa_hip.so.

PyTorch provides device memory (caching allocator), the current HIP stream and
the autograd tape only; every arithmetic op below runs in a hand-written gfx950
kernel.  There is deliberately no CPU / eager-PyTorch fallback: CPU tensors or a
missing library raise.
"""
import ctypes
import weakref

import torch

from . import _lib as L
from ._lib import ConvDesc

ACT_NONE, ACT_RELU, ACT_LRELU, ACT_SIGMOID, ACT_ELU, ACT_SELU = 0, 1, 2, 3, 4, 5   # (SELU: activation() only, never fused)
BN_EPS = 1e-5
LN_EPS = 1e-5


# --------------------------------------------------------------------------- plumbing
def _lib():
    return L.load()


def _c(t, name="tensor"):
    """contiguous fp32 HIP tensor or raise (no fallback)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"multipitch_architectures_amd: {name} lives on {t.device}; the HIP kernels need a "
                           "GPU tensor and there is no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"multipitch_architectures_amd: {name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(rc, what):
    if rc != 0:
        L.check(rc, what)


# parameter epoch: bumped by the fused optimizer (which writes parameters through raw pointers)
_param_epoch = 0
_pack_cache = {}


def bump_param_epoch():
    global _param_epoch
    _param_epoch += 1


class _Rng:
    """Counter-based dropout stream.  Element i of a dropout call draws from position base + local + i of the stream
    `seed`; seed and base live in device memory (so a captured HIP graph of the training step draws fresh masks at every
    replay), `local` is the call's position inside the current step and is a plain kernel argument."""
    seed = 0x5EED
    local = 0            # elements drawn since the last rng_advance() (host mirror; static inside a captured step)
    state = {}           # device -> int64[2] tensor: [seed, base]


def manual_seed(seed: int):
    _Rng.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    _Rng.local = 0
    _Rng.state.clear()


def _rng_state(device):
    st = _Rng.state.get(device)
    if st is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the dropout stream state must exist before a HIP graph is captured: run one eager "
                               "(warm-up) step first")
        seed = _Rng.seed - (1 << 64) if _Rng.seed >= (1 << 63) else _Rng.seed
        st = torch.tensor([seed, 0], dtype=torch.int64, device=device)
        _Rng.state[device] = st
    return st


_seed_ones = {}


def backward_seed(loss):
    """d loss / d loss = 1 as a persistent tensor: `loss.backward(ops.backward_seed(loss))` spares the ones_like fill kernel
    autograd launches for a bare `loss.backward()` (the last ATen kernel of a training step)."""
    key = (loss.device, loss.dtype, tuple(loss.shape))
    t = _seed_ones.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the backward seed must exist before a HIP graph is captured: run one eager step first")
        t = _seed_ones[key] = torch.ones(loss.shape, dtype=loss.dtype, device=loss.device)
    return t


def rng_advance():
    """End of a training step: move the device-side base of the dropout stream past everything the step drew.  One tiny
    kernel; inside a captured graph it is what makes every replay use new random numbers."""
    if _Rng.local:
        for st in _Rng.state.values():
            _chk(_lib().mpa_u64_add(ctypes.c_void_p(st.data_ptr() + 8), _Rng.local, _s()), "mpa_u64_add")
    _Rng.local = 0


def _packed(weight, desc, mode):
    """Packed filter bank for (weight, problem geometry, fwd/bwd-data), cached until the weight changes."""
    sig = (weight.data_ptr(), weight._version, _param_epoch)
    ent = _pack_cache.get(id(weight))
    if ent is None or ent[0]() is not weight or ent[1] != sig:
        ent = (weakref.ref(weight), sig, {})
        _pack_cache[id(weight)] = ent
        if len(_pack_cache) > 1024:
            for k in [k for k, v in _pack_cache.items() if v[0]() is None]:
                del _pack_cache[k]
    key = (desc.key(), mode)          # the tiling (hence the packed layout) depends on the batch size too
    buf = ent[2].get(key)
    if buf is None:
        lib = _lib()
        n = lib.mpa_conv2d_packed_floats(ctypes.byref(desc), mode)
        if n < 0:
            L.check(int(n), "mpa_conv2d_packed_floats")
        # a bank that was packed for an earlier state of this weight is re-used (same size, same place): the table of
        # pack tables (run_pack_table) point at it
        buf = _pack_buffers.get((id(weight), key))
        if buf is None or buf[0]() is not weight or buf[1].numel() != int(n):
            if buf is not None:
                _pack_keepalive.append(buf[1])
            buf = (weakref.ref(weight), torch.empty(int(n), dtype=torch.float32, device=weight.device),
                   ctypes.string_at(ctypes.byref(desc), ctypes.sizeof(desc)), mode)
            _pack_buffers[(id(weight), key)] = buf
        buf = buf[1]
        _chk(lib.mpa_conv2d_pack(ctypes.byref(desc), mode, _p(weight), _p(buf), _s()), "mpa_conv2d_pack")
        ent[2][key] = buf
    _pack_used.add((id(weight), key))
    return buf


_pack_buffers = {}      # (id(weight), key) -> (weakref(weight), packed buffer, raw conv desc, mode): stable storage
_pack_used = set()      # banks asked for since begin_pack_window()


class PackTable:
    """device table of filter banks (entries of mpa_conv2d_pack_entry): one launch re-packs all of them"""

    def __init__(self, keys, device, n, wptrs=()):
        self.keys, self.device, self.n = keys, device, n
        self.wptrs = wptrs          # data_ptr() of each key's weight when the table was built (baked into the entries)


def pack_table_valid(tab):
    """every weight of the table is alive and still at the address the table's entries carry"""
    for (wid, key), ptr in zip(tab.keys, tab.wptrs):
        ent = _pack_buffers.get((wid, key))
        w = ent[0]() if ent is not None else None
        if w is None or w.data_ptr() != ptr:
            return False
    return True


_pack_keepalive = []      # replaced banks: a captured graph may still launch on them


def begin_pack_window():
    """forget which banks were used so far: pack_window_keys() then names exactly the banks used from here on (a training
    step calls this first, so that evaluation passes between steps do not leak their shapes into the step's table)"""
    global _pack_used
    _pack_used = set()


def pack_window_keys():
    return tuple(sorted(k for k in _pack_used if k in _pack_buffers and _pack_buffers[k][0]() is not None))


def build_pack_table(keys):
    """table for run_pack_table(): the banks `keys` (from pack_window_keys()) with their weights and geometries.  Copies
    a few KB to the device -- not inside a graph capture."""
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("the set of filter banks changed inside a graph capture: run one eager step first")
    if not keys:
        return PackTable((), None, 0)
    for k in [k for k, v in _pack_buffers.items() if v[0]() is None]:      # banks of weights that no longer exist
        del _pack_buffers[k]
    lib = _lib()
    esz = lib.mpa_conv2d_pack_entry_bytes()
    host = ctypes.create_string_buffer(esz * len(keys))
    dev = None
    for i, k in enumerate(keys):
        wref, buf, raw, mode = _pack_buffers[k]
        w = wref()
        desc = L.ConvDesc.from_buffer_copy(raw)
        _chk(lib.mpa_conv2d_pack_entry(ctypes.byref(desc), mode, _p(w), _p(buf), ctypes.byref(host, i * esz)),
             "mpa_conv2d_pack_entry")
        dev = w.device
    return PackTable(keys, torch.frombuffer(bytearray(host.raw), dtype=torch.uint8).to(dev), len(keys),
                     tuple(_pack_buffers[k][0]().data_ptr() for k in keys))


def run_pack_table(tab):
    """Re-pack, in one launch, every filter bank of the table for the weights as they are now (a training step does this
    first: its forward / backward then finds every bank current instead of re-packing 2 banks per convolution lazily,
    44 small launches for SAUnet:L).  Banks of the same weights that are not in the table (other batch sizes, evaluation
    shapes) are dropped from the cache and re-packed lazily when they are needed again."""
    if tab.n == 0:
        return
    if not pack_table_valid(tab):
        raise RuntimeError("a weight of this pack table no longer exists or has moved: build a new table")
    _chk(_lib().mpa_conv2d_pack_many(_p(tab.device), tab.n, _s()), "mpa_conv2d_pack_many")
    per_weight = {}
    for wid, key in tab.keys:
        per_weight.setdefault(wid, []).append(key)
    for wid, keys in per_weight.items():
        w = _pack_buffers[(wid, keys[0])][0]()
        sig = (w.data_ptr(), w._version, _param_epoch)
        _pack_cache[wid] = (weakref.ref(w), sig, {k: _pack_buffers[(wid, k)][1] for k in keys})


# --------------------------------------------------------------------------- data-parallel exactness modes
class _Sync:
    bn = False            # BatchNorm statistics over the *global* batch: all-reduce of 2*C sums per layer and direction
    attn = False          # batch-axis attention over the *global* batch: all-gather of K and V, reduce of dK and dV
    group = None


def set_data_parallel_exactness(sync_bn=False, gather_attention=False, group=None):
    """By default a data-parallel rank normalises and attends over its local shard (what DistributedDataParallel around
    the reference would do), so an N x B/N run computes a slightly different model than the 1 x B run.  With these modes
    on, train-mode BatchNorm uses the statistics of the whole batch (SyncBN: the per-channel sums are all-reduced between
    the two halves of each pass) and the batch-axis attention sees the keys / values of every rank -- an N x B/N run then
    reproduces the 1 x B run's loss and gradients to fp32 rounding.  Both put collectives inside forward / backward:
    `step.TrainStep` launches such steps kernel by kernel (no HIP graph)."""
    _Sync.bn, _Sync.attn, _Sync.group = bool(sync_bn), bool(gather_attention), group


def _dp_world():
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(_Sync.group)


def exactness_collectives_active():
    return (_Sync.bn or _Sync.attn) and _dp_world() > 1


# --------------------------------------------------------------------------- split-bf16 ("bf16x3") convolution path
class _Precision:
    conv = "f32"          # "f32": exact fp32-input MFMA (default, the headline) | "bf16x3": hi/lo bf16 split, 3 MFMAs, fp32 acc


def set_conv_precision(mode: str):
    """Arithmetic of the convolutions that have a split-bf16 kernel (15-row filters, stride 1): "f32" (default) or
    "bf16x3" -- every operand as hi + lo bf16 halves, products hi*hi + hi*lo + lo*hi accumulated in fp32.  The
    bf16x3 path agrees with the reference to ~2e-5 of a layer output's rms (inside the 1e-4 forward bound) but is
    not bit-identical to the exact path; layers without a bf16x3 kernel keep the exact one.

    Reproducibility: the exact path reduces every split sum in a fixed order (run-to-run bit-identical but for the
    backward-data launches that add channel slices atomically at small batches); in this mode the weight-gradient GEMMs
    of the linear / LSTM layers (`mpa_gemm_bf16x3`) additionally split K over workgroups that add into C atomically, so
    two runs of the same step agree to rounding, not bit for bit."""
    if mode not in ("f32", "bf16x3"):
        raise ValueError(f"conv precision must be 'f32' or 'bf16x3', got {mode!r}")
    _Precision.conv = mode


def get_conv_precision():
    return _Precision.conv


def _bfx_ok(desc, mode):
    return _Precision.conv == "bf16x3" and bool(_lib().mpa_conv2d_bf16x3_supported(ctypes.byref(desc), mode))


def split_bf16(x):
    """fp32 (B,C,H,W) -> hi/lo bf16 planes [B][ceil(C/8)][2][H][W][8] (an int32 tensor of 4 words per 8-channel granule)"""
    x = _c(x, "split input")
    B, C, H, W = x.shape
    out = torch.empty((B, (C + 7) // 8, 2, H, W, 4), dtype=torch.int32, device=x.device)
    _chk(_lib().mpa_bf16x3_split(_p(x), _p(out), B, C, H, W, _s()), "mpa_bf16x3_split")
    return out


def _packed_bfx(weight, desc, mode):
    """bf16x3 filter bank (mode 0 forward, 1 backward-data), cached like the fp32 banks until the weight changes"""
    sig = (weight.data_ptr(), weight._version, _param_epoch)
    ent = _bfx_cache.get(id(weight))
    if ent is None or ent[0]() is not weight or ent[1] != sig:
        # every bank this weight ever had stays alive (and in place) as long as the weight does: a captured graph packs
        # into and reads the banks of *its* shapes, whichever banks the passes in between (evaluation: forward only) used
        old = {**ent[3], **ent[2]} if ent is not None and ent[0]() is weight else {}
        ent = (weakref.ref(weight), sig, {}, old)
        _bfx_cache[id(weight)] = ent
        if len(_bfx_cache) > 1024:
            for k in [k for k, v in _bfx_cache.items() if v[0]() is None]:
                del _bfx_cache[k]
    key = (desc.key()[1:], mode)         # the bank does not depend on the batch size
    buf = ent[2].get(key)
    if buf is None:
        lib = _lib()
        n = lib.mpa_conv2d_bf16x3_packed_bytes(ctypes.byref(desc), mode)
        if n < 0:
            L.check(int(n), "mpa_conv2d_bf16x3_packed_bytes")
        buf = ent[3].get(key)            # same storage as for the previous weight version (a captured graph packs into it)
        if buf is None or buf.numel() * 4 != int(n):
            buf = torch.empty(int(n) // 4, dtype=torch.int32, device=weight.device)
        _chk(lib.mpa_conv2d_bf16x3_pack(ctypes.byref(desc), mode, _p(weight), _p(buf), _s()), "mpa_conv2d_bf16x3_pack")
        ent[2][key] = buf
    return buf


_bfx_cache = {}


# optional HIP-event probe around one class of conv launches (bench.py's live roofline measurement)
class _Probe:
    match = None        # callable(desc_key, kind) -> bool
    records = []        # (start_event, end_event)


def set_kernel_probe(match):
    _Probe.match = match
    _Probe.records = []


def probe_results_ms():
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in _Probe.records]


def _probed(kind, desc, fn):
    if _Probe.match is not None and _Probe.match(desc.key(), kind):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn()
        b.record()
        _Probe.records.append((a, b))
        return rc
    return fn()


# --------------------------------------------------------------------------- convolution
class Conv2dFn(torch.autograd.Function):
    """nn.Conv2d.  with_stats: the convolution sits in front of a BatchNorm2d (double_conv, unet_cnns.py:49-59) and its
    store epilogue also leaves per-(pixel tile, channel) partial sums of y and y^2 -- returned as a second,
    non-differentiable output that `batchnorm_relu(..., partials=...)` turns into the batch statistics without another
    pass over y."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, act, slope, with_stats=False, act_bwd_by_consumer=False):
        # act_bwd_by_consumer: the activation is applied here (store epilogue) but its backward pass by the node that consumes y
        # (poolrows_dropout_add(..., producer_slope=...)): backward() then takes the incoming gradient as d/d(pre-activation)
        ctx.defer_act = bool(act_bwd_by_consumer)
        # no zero tensor for the gradient of the non-differentiable `partials` output: autograd would otherwise fill one per
        # BatchNorm layer and step (18 fill launches for SAUnet:L)
        ctx.set_materialize_grads(False)
        x, weight, bias = _c(x, "conv input"), _c(weight, "conv weight"), _c(bias, "conv bias")
        B, Cin, H, W = x.shape
        Cout, Cin_w, kh, kw = weight.shape
        if Cin_w != Cin:
            raise RuntimeError(f"conv2d: input has {Cin} channels, weight expects {Cin_w}")
        d = ConvDesc(B, Cin, H, W, Cout, kh, kw, stride[0], stride[1], padding[0], padding[1])
        if d.OH <= 0 or d.OW <= 0:
            raise RuntimeError(f"conv2d: kernel {(kh, kw)} larger than padded input {(H, W)}")
        y = torch.empty((B, Cout, d.OH, d.OW), dtype=torch.float32, device=x.device)
        ctx.desc, ctx.act, ctx.slope, ctx.has_bias = d, act, float(slope), bias is not None
        ctx.bfx_xs = False
        if _bfx_ok(d, 0):
            lib = _lib()
            xs, wpb = split_bf16(x), _packed_bfx(weight, d, 0)
            # the backward-weight kernel reads the split input: keep that instead of x (same number of bytes)
            # ... unless the backward-data pass of this geometry has no bf16x3 kernel (pw > kw - 1): it then runs on the exact
            # kernel, which reads x
            ctx.bfx_xs = bool(lib.mpa_conv2d_bf16x3_supported(ctypes.byref(d), 2)) and \
                (not ctx.needs_input_grad[0] or bool(lib.mpa_conv2d_bf16x3_supported(ctypes.byref(d), 1)))
            ctx.save_for_backward(xs if ctx.bfx_xs else x, weight, y if act != ACT_NONE else None)
            partials = None
            if with_stats:
                if act != ACT_NONE:
                    raise RuntimeError("conv2d(with_stats=True) is the convolution in front of a BatchNorm: no activation")
                rows = lib.mpa_conv2d_bf16x3_stats_rows(ctypes.byref(d))
                if rows < 0:
                    L.check(int(rows), "mpa_conv2d_bf16x3_stats_rows")
                partials = torch.empty((int(rows), Cout, 2), dtype=torch.float32, device=x.device)
            _chk(_probed("fwd", d, lambda: lib.mpa_conv2d_bf16x3_fwd(ctypes.byref(d), _p(xs), _p(wpb), _p(bias), _p(y), act,
                                                                    float(slope), _p(partials), _s())),
                 "mpa_conv2d_bf16x3_fwd")
            if with_stats:
                ctx.mark_non_differentiable(partials)
                return y, partials
            return y
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        if not with_stats and _lib().mpa_conv2d_fold_supported(ctypes.byref(d)):
            # 15x15 layer with a cout remainder (70 = 64 + 6): two launches, the remainder as row pairs of one MFMA tile
            wp = _packed(weight, d, 2)
            _chk(_probed("fwd", d, lambda: _lib().mpa_conv2d_fwd_folded(ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(y), act,
                                                                        float(slope), _s())), "mpa_conv2d_fwd_folded")
            return y
        wp = _packed(weight, d, 0)
        if with_stats:
            if act != ACT_NONE:
                raise RuntimeError("conv2d(with_stats=True) is the convolution in front of a BatchNorm: no activation")
            rows = _lib().mpa_conv2d_fwd_stats_rows(ctypes.byref(d))
            if rows < 0:
                L.check(int(rows), "mpa_conv2d_fwd_stats_rows")
            partials = torch.empty((int(rows), Cout, 2), dtype=torch.float32, device=x.device)
            _chk(_probed("fwd", d, lambda: _lib().mpa_conv2d_fwd_stats(ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(y),
                                                                       _p(partials), _s())), "mpa_conv2d_fwd_stats")
            ctx.mark_non_differentiable(partials)
            return y, partials
        _chk(_probed("fwd", d, lambda: _lib().mpa_conv2d_fwd(ctypes.byref(d), _p(x), _p(wp), _p(bias), _p(y), act,
                                                             float(slope), _s())), "mpa_conv2d_fwd")
        return y

    @staticmethod
    def backward(ctx, dy, _dpartials=None):
        if dy is None:                     # (set_materialize_grads(False): nothing flowed into y)
            return None, None, None, None, None, None, None, None, None
        x, weight, y = ctx.saved_tensors
        d, lib = ctx.desc, _lib()
        dy = _c(dy, "conv grad")
        if ctx.act != ACT_NONE and not ctx.defer_act:
            g = torch.empty_like(dy)
            _chk(lib.mpa_act_bwd(_p(dy), _p(y), _p(g), dy.numel(), ctx.act, ctx.slope, _s()), "mpa_act_bwd")
            dy = g
        dx = dw = db = None
        dys = None
        x_shape = (d.B, d.Cin, d.H, d.W)
        if ctx.needs_input_grad[0] and _bfx_ok(d, 1):
            dx = torch.empty(x_shape, dtype=torch.float32, device=dy.device)
            dys, wpb = split_bf16(dy), _packed_bfx(weight, d, 1)
            _chk(_probed("dgrad", d, lambda: lib.mpa_conv2d_bf16x3_bwd_data(ctypes.byref(d), _p(dys), _p(wpb), _p(dx),
                                                                           _s())), "mpa_conv2d_bf16x3_bwd_data")
        elif ctx.needs_input_grad[0]:
            if ctx.bfx_xs:
                raise RuntimeError("conv precision changed between the forward and the backward pass of a convolution")
            dx = torch.empty_like(x)
            wp = _packed(weight, d, 1)
            _chk(_probed("dgrad", d, lambda: lib.mpa_conv2d_bwd_data(ctypes.byref(d), _p(dy), _p(wp), _p(dx), _s())),
                 "mpa_conv2d_bwd_data")
        if (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])) and ctx.bfx_xs:
            dw = torch.empty_like(weight)
            db = torch.empty(d.Cout, dtype=torch.float32, device=dy.device) if ctx.has_bias else None
            if dys is None:
                dys = split_bf16(dy)
            nbytes = lib.mpa_conv2d_bf16x3_bwd_weight_workspace(ctypes.byref(d))
            if nbytes < 0:
                L.check(int(nbytes), "mpa_conv2d_bf16x3_bwd_weight_workspace")
            ws = torch.empty(int(nbytes) // 4, dtype=torch.float32, device=dy.device)
            _chk(_probed("wgrad", d, lambda: lib.mpa_conv2d_bf16x3_bwd_weight(ctypes.byref(d), _p(x), _p(dys), _p(dw), _p(db),
                                                                              _p(ws), int(nbytes), _s())),
                 "mpa_conv2d_bf16x3_bwd_weight")
        elif ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw = torch.empty_like(weight)
            db = torch.empty(d.Cout, dtype=torch.float32, device=x.device) if ctx.has_bias else None
            nbytes = lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d))
            if nbytes < 0:
                L.check(int(nbytes), "mpa_conv2d_bwd_weight_workspace")
            ws = torch.empty(int(nbytes) // 4, dtype=torch.float32, device=x.device)
            _chk(_probed("wgrad", d, lambda: lib.mpa_conv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(db),
                                                                       _p(ws), int(nbytes), _s())),
                 "mpa_conv2d_bwd_weight")
        return dx, dw, db, None, None, None, None, None, None


def conv2d_stats(x, weight, bias, stride=(1, 1), padding=(0, 0)):
    """(y, partials): convolution + the partial sums its BatchNorm needs (see Conv2dFn)"""
    return Conv2dFn.apply(x, weight, bias, tuple(stride), tuple(padding), ACT_NONE, 0.0, True)


def conv2d(x, weight, bias, stride=(1, 1), padding=(0, 0), act=ACT_NONE, slope=0.0, act_bwd_by_consumer=False):
    """act_bwd_by_consumer: see Conv2dFn.forward (only the plain convolution path takes it: the caller pairs it with
    poolrows_dropout_add(..., producer_slope=...), whose stages are padded (kh,kw) convolutions)."""
    stride, padding = tuple(stride), tuple(padding)
    kh, kw = weight.shape[2], weight.shape[3]
    if act_bwd_by_consumer:
        if act not in (ACT_RELU, ACT_LRELU) or (kh > 1 and kh == x.shape[2] and padding[0] == 0):
            raise RuntimeError("conv2d(act_bwd_by_consumer=True): a ReLU / LeakyReLU stage on the plain convolution path only")
        return Conv2dFn.apply(x, weight, bias, stride, padding, act, slope, False, True)
    if kh > 1 and kh == x.shape[2] and padding[0] == 0 and x.is_contiguous() and weight.is_contiguous():
        B, C, H, W = x.shape
        if kw == 1 and stride[1] == 1 and padding[1] == 0:
            # full-height valid (kh,1) kernel -- conv3's (75,1) on a 75-frame patch (unet_cnns.py:545, basic_cnns.py:178):
            # the output is one row per sample, y[b,co,0,w] = sum_k W2[co,k] X[b,k,w] with k = (ci,dy) adjacent in NCHW,
            # i.e. a plain GEMM over the B*W output positions.  As a convolution it would run one 72-pixel tile per
            # sample through 6000 input channels with a barrier per 32-channel chunk (3.5 ms at batch 256); as a GEMM
            # (one transposing copy of x, 128-row tiles, split-K) it is ~10x faster.
            x2 = transpose_last2(x.view(B, C * H, W))                      # (B, W, C*H)
            y2 = linear(x2, weight.view(weight.shape[0], C * kh), bias)    # (B, W, Cout)
            if act != ACT_NONE:
                y2 = activation(y2, act, slope)
            return transpose_last2(y2).view(B, weight.shape[0], 1, W)
        # other full-height valid kernels: a 1 x kw conv over Cin*kh channels -- pure views, no data movement
        y = Conv2dFn.apply(x.view(B, C * H, 1, W), weight.view(weight.shape[0], C * kh, 1, kw), bias,
                           (1, stride[1]), (0, padding[1]), act, slope)
        return y
    return Conv2dFn.apply(x, weight, bias, stride, padding, act, slope)


# --------------------------------------------------------------------------- normalisation
class LayerNormCFFn(torch.autograd.Function):
    """LayerNorm([C,F]) on x.transpose(1,2) -- unet_cnns.py:560."""

    @staticmethod
    def forward(ctx, x, w, b):
        x, w, b = _c(x, "input"), _c(w), _c(b)
        B, C, T, F = x.shape
        y = torch.empty_like(x)
        mean = torch.empty(B * T, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        _chk(_lib().mpa_layernorm_cf_fwd(_p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), B, C, T, F, LN_EPS, _s()),
             "mpa_layernorm_cf_fwd")
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        B, C, T, F = x.shape
        lib = _lib()
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw, db = torch.empty_like(w), torch.empty_like(w)
        ws = torch.empty(lib.mpa_layernorm_bwd_workspace(C * F) // 4, dtype=torch.float32, device=x.device)
        _chk(lib.mpa_layernorm_cf_bwd_ws(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(dw), _p(db), _p(ws),
                                        B, C, T, F, _s()), "mpa_layernorm_cf_bwd_ws")
        return dx, dw, db


def layernorm_cf(x, w, b):
    return LayerNormCFFn.apply(x, w, b)


class LayerNormRowsFn(torch.autograd.Function):
    """y = LayerNorm_E(a + r) over the last dim (unet_cnns.py:156,158); r may be None."""

    @staticmethod
    def forward(ctx, a, r, w, b):
        a, r, w, b = _c(a), _c(r), _c(w), _c(b)
        E = a.shape[-1]
        rows = a.numel() // E
        y = torch.empty_like(a)
        xs = torch.empty_like(a) if r is not None else a
        mean = torch.empty(rows, dtype=torch.float32, device=a.device)
        rstd = torch.empty_like(mean)
        _chk(_lib().mpa_layernorm_rows_fwd(_p(a), _p(r), _p(w), _p(b), _p(xs) if r is not None else None, _p(y),
                                          _p(mean), _p(rstd), rows, E, LN_EPS, _s()), "mpa_layernorm_rows_fwd")
        ctx.has_r = r is not None
        ctx.save_for_backward(xs, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, w, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        E = xs.shape[-1]
        rows = xs.numel() // E
        lib = _lib()
        dx = torch.empty_like(xs)
        dw, db = torch.empty_like(w), torch.empty_like(w)
        ws = torch.empty(lib.mpa_layernorm_bwd_workspace(E) // 4, dtype=torch.float32, device=xs.device)
        _chk(lib.mpa_layernorm_rows_bwd_ws(_p(dy), _p(xs), _p(w), _p(mean), _p(rstd), _p(dx), _p(dw), _p(db), _p(ws),
                                          rows, E, _s()), "mpa_layernorm_rows_bwd_ws")
        return dx, (dx if ctx.has_r else None), dw, db


def layernorm_rows(a, r, w, b):
    return LayerNormRowsFn.apply(a, r, w, b)


class BatchNormReLUFn(torch.autograd.Function):
    """nn.BatchNorm2d (+ fused nn.ReLU) of double_conv -- unet_cnns.py:51-52."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, nbt, training, momentum, relu, partials=None):
        x, gamma, beta = _c(x), _c(gamma), _c(beta)
        B, C, H, W = x.shape
        y = torch.empty_like(x)
        save_mean = torch.empty(C, dtype=torch.float32, device=x.device)
        save_invstd = torch.empty_like(save_mean)
        lib = _lib()
        ctx.sync_world = 1
        if training and _Sync.bn and _dp_world() > 1:
            # SyncBN: this rank's sums, all-reduced, then normalisation with the global count (equal shards)
            import torch.distributed as dist
            world = _dp_world()
            sums = torch.empty(2 * C, dtype=torch.float64, device=x.device)
            _chk(lib.mpa_bn_batch_sums(_p(x), ctypes.c_void_p(sums.data_ptr()), B, C, H * W, _s()), "mpa_bn_batch_sums")
            dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=_Sync.group)
            _chk(lib.mpa_bn_relu_train_fwd_sums(_p(x), ctypes.c_void_p(sums.data_ptr()), float(B * H * W * world), _p(gamma),
                                               _p(beta), _p(running_mean), _p(running_var),
                                               ctypes.c_void_p(nbt.data_ptr()) if nbt is not None else None, _p(y),
                                               _p(save_mean), _p(save_invstd), B, C, H * W, float(momentum), BN_EPS,
                                               int(relu), _s()), "mpa_bn_relu_train_fwd_sums")
            ctx.sync_world = world
        elif training and partials is not None:
            # batch statistics from the producing convolution's epilogue: no statistics pass over x
            if partials.shape[1:] != (C, 2):
                raise RuntimeError(f"batchnorm: partial sums of shape {tuple(partials.shape)} for {C} channels")
            ws = torch.empty(64 * C * 2, dtype=torch.float64, device=x.device) if partials.shape[0] > 512 else None
            _chk(lib.mpa_bn_relu_train_fwd_partials(_p(x), _p(partials), partials.shape[0], _p(gamma), _p(beta),
                                                   _p(running_mean), _p(running_var),
                                                   ctypes.c_void_p(nbt.data_ptr()) if nbt is not None else None, _p(y),
                                                   _p(save_mean), _p(save_invstd),
                                                   ctypes.c_void_p(ws.data_ptr()) if ws is not None else None, B, C,
                                                   H * W, float(momentum), BN_EPS, int(relu), _s()),
                 "mpa_bn_relu_train_fwd_partials")
        elif training:
            ws = torch.empty(2 * C, dtype=torch.float64, device=x.device)
            _chk(lib.mpa_bn_relu_train_fwd(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                          ctypes.c_void_p(nbt.data_ptr()) if nbt is not None else None, _p(y),
                                          _p(save_mean), _p(save_invstd), ctypes.c_void_p(ws.data_ptr()), B, C, H * W,
                                          float(momentum), BN_EPS, int(relu), _s()), "mpa_bn_relu_train_fwd")
        else:
            _chk(lib.mpa_bn_relu_eval_fwd(_p(x), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(y),
                                         _p(save_mean), _p(save_invstd), B, C, H * W, BN_EPS, int(relu), _s()),
                 "mpa_bn_relu_eval_fwd")
        ctx.training, ctx.relu = bool(training), bool(relu)
        # the backward recomputes the ReLU mask from x (same arithmetic as the forward), so y is not kept alive here
        ctx.save_for_backward(x, gamma, beta, save_mean, save_invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, beta, save_mean, save_invstd = ctx.saved_tensors
        dy = _c(dy)
        B, C, H, W = x.shape
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        ws = torch.empty(128 * C, dtype=torch.float64, device=x.device)       # (2 C for the SyncBN halves)
        if ctx.training and ctx.sync_world > 1:
            import torch.distributed as dist
            lib = _lib()
            _chk(lib.mpa_bn_relu_bwd_sums(_p(dy), _p(x), _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
                                         ctypes.c_void_p(ws.data_ptr()), _p(dgamma), _p(dbeta), B, C, H * W, int(ctx.relu),
                                         _s()), "mpa_bn_relu_bwd_sums")
            dist.all_reduce(ws, op=dist.ReduceOp.SUM, group=_Sync.group)
            _chk(lib.mpa_bn_relu_bwd_apply(_p(dy), _p(x), _p(gamma), _p(beta), _p(save_mean), _p(save_invstd),
                                          ctypes.c_void_p(ws.data_ptr()), float(B * H * W * ctx.sync_world), _p(dx), B, C,
                                          H * W, int(ctx.relu), _s()), "mpa_bn_relu_bwd_apply")
            return dx, dgamma, dbeta, None, None, None, None, None, None, None
        _chk(_lib().mpa_bn_relu_bwd(_p(dy), _p(x), None, _p(gamma), _p(beta), _p(save_mean), _p(save_invstd), _p(dx), _p(dgamma),
                                   _p(dbeta), ctypes.c_void_p(ws.data_ptr()), B, C, H * W, int(ctx.relu),
                                   int(ctx.training), _s()), "mpa_bn_relu_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


def batchnorm_relu(x, gamma, beta, running_mean, running_var, nbt, training, momentum=0.1, relu=True, partials=None):
    return BatchNormReLUFn.apply(x, gamma, beta, running_mean, running_var, nbt, training, momentum, relu, partials)


# --------------------------------------------------------------------------- pooling / upsampling
class MaxPool2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        x = _c(x)
        B, C, H, W = x.shape
        OH = (H + 2 * p[0] - k[0]) // s[0] + 1
        OW = (W + 2 * p[1] - k[1]) // s[1] + 1
        if OH <= 0 or OW <= 0:
            raise RuntimeError(f"max_pool2d: window {k} larger than input {(H, W)}")
        y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
        need = ctx.needs_input_grad[0]
        idx = torch.empty((B, C, OH, OW), dtype=torch.int32, device=x.device) if need else None
        _chk(_lib().mpa_maxpool2d_fwd(_p(x), _p(y), ctypes.c_void_p(idx.data_ptr()) if need else None, B, C, H, W,
                                     k[0], k[1], s[0], s[1], p[0], p[1], _s()), "mpa_maxpool2d_fwd")
        ctx.geom = (B, C, H, W, k, s, p)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, H, W, k, s, p = ctx.geom
        dy = _c(dy)
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dy.device)
        _chk(_lib().mpa_maxpool2d_bwd(_p(dy), ctypes.c_void_p(idx.data_ptr()), _p(dx), B, C, H, W, k[0], k[1], s[0],
                                     s[1], p[0], p[1], _s()), "mpa_maxpool2d_bwd")
        return dx, None, None, None


def max_pool2d(x, kernel, stride=None, padding=(0, 0)):
    stride = kernel if stride is None else stride
    return MaxPool2dFn.apply(x, tuple(kernel), tuple(stride), tuple(padding))


class MaxPoolIdxFn(torch.autograd.Function):
    """nn.MaxPool2d(kernel, return_indices=True) with stride == kernel (unet_cnns.py:1715-1729): (y, indices); the indices
    (flat offsets inside each H*W plane, int32 here, int64 upstream) feed MaxUnpoolFn and are not differentiable."""

    @staticmethod
    def forward(ctx, x, k):
        x = _c(x)
        B, C, H, W = x.shape
        OH, OW = H // k[0], W // k[1]
        if OH <= 0 or OW <= 0:
            raise RuntimeError(f"max_pool2d: window {k} larger than input {(H, W)}")
        y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
        idx = torch.empty((B, C, OH, OW), dtype=torch.int32, device=x.device)
        _chk(_lib().mpa_maxpool2d_fwd(_p(x), _p(y), ctypes.c_void_p(idx.data_ptr()), B, C, H, W, k[0], k[1], k[0], k[1], 0, 0,
                                     _s()), "mpa_maxpool2d_fwd")
        ctx.geom = (B, C, H, W, k)
        ctx.save_for_backward(idx)
        ctx.mark_non_differentiable(idx)
        return y, idx

    @staticmethod
    def backward(ctx, dy, _didx):
        (idx,) = ctx.saved_tensors
        B, C, H, W, k = ctx.geom
        dy = _c(dy)
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dy.device)
        _chk(_lib().mpa_maxpool2d_bwd(_p(dy), ctypes.c_void_p(idx.data_ptr()), _p(dx), B, C, H, W, k[0], k[1], k[0], k[1], 0, 0,
                                     _s()), "mpa_maxpool2d_bwd")
        return dx, None


def max_pool2d_with_indices(x, kernel):
    return MaxPoolIdxFn.apply(x, tuple(kernel))


class MaxUnpoolFn(torch.autograd.Function):
    """nn.MaxUnpool2d(kernel)(x, indices) for the indices of max_pool2d_with_indices (stride == kernel)."""

    @staticmethod
    def forward(ctx, x, idx, k):
        x = _c(x)
        B, C, OH, OW = x.shape
        if idx.dtype != torch.int32 or tuple(idx.shape) != (B, C, OH, OW) or not idx.is_contiguous():
            raise RuntimeError("max_unpool2d: indices must be the contiguous int32 tensor max_pool2d_with_indices returned")
        y = torch.empty((B, C, OH * k[0], OW * k[1]), dtype=torch.float32, device=x.device)
        _chk(_lib().mpa_maxunpool2d_fwd(_p(x), ctypes.c_void_p(idx.data_ptr()), _p(y), B, C, OH, OW, k[0], k[1], _s()),
             "mpa_maxunpool2d_fwd")
        ctx.k = k
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        (idx,) = ctx.saved_tensors
        B, C, OH, OW = idx.shape
        dy = _c(dy)
        dx = torch.empty((B, C, OH, OW), dtype=torch.float32, device=dy.device)
        _chk(_lib().mpa_maxunpool2d_bwd(_p(dy), ctypes.c_void_p(idx.data_ptr()), _p(dx), B, C, OH, OW, ctx.k[0], ctx.k[1], _s()),
             "mpa_maxunpool2d_bwd")
        return dx, None, None


def max_unpool2d(x, indices, kernel):
    return MaxUnpoolFn.apply(x, indices, tuple(kernel))


class PoolSkipFn(torch.autograd.Function):
    """x -> (max_pool2d(x, k, s, p), x): an encoder level's output feeds the next level's MaxPool2d *and* the decoder's
    unet_up_concat_padding (unet_cnns.py:562-571: x1 .. x4).  As one autograd node the two gradients meet here instead of
    in autograd's accumulation (an ATen add kernel): the pool backward starts its LDS plane from the skip gradient -- read
    in place from the first channels of the concatenated gradient (UpCatFn with skip_view) -- and writes their sum, one
    pass instead of pool backward + slice copy + add."""

    @staticmethod
    def forward(ctx, x, k, s, p):
        ctx.set_materialize_grads(False)
        x = _c(x)
        B, C, H, W = x.shape
        OH = (H + 2 * p[0] - k[0]) // s[0] + 1
        OW = (W + 2 * p[1] - k[1]) // s[1] + 1
        if OH <= 0 or OW <= 0:
            raise RuntimeError(f"max_pool2d: window {k} larger than input {(H, W)}")
        y = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
        need = ctx.needs_input_grad[0]
        idx = torch.empty((B, C, OH, OW), dtype=torch.int32, device=x.device) if need else None
        _chk(_lib().mpa_maxpool2d_fwd(_p(x), _p(y), ctypes.c_void_p(idx.data_ptr()) if need else None, B, C, H, W,
                                     k[0], k[1], s[0], s[1], p[0], p[1], _s()), "mpa_maxpool2d_fwd")
        ctx.geom = (B, C, H, W, k, s, p)
        ctx.save_for_backward(idx)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dskip):
        B, C, H, W, k, s, p = ctx.geom
        if dy is None:                                  # only the skip path was used
            return dskip, None, None, None
        (idx,) = ctx.saved_tensors
        dy = _c(dy)
        add, add_bs = None, 0
        if dskip is not None:
            if dskip.dtype != torch.float32 or not dskip.is_cuda or dskip.shape != (B, C, H, W):
                raise RuntimeError("pool_skip: unexpected skip gradient")
            st = dskip.stride()
            if C * H * W == 0 or not (st[3] == 1 and st[2] == W and st[1] == H * W and st[0] >= C * H * W):
                dskip = dskip.contiguous()              # (never on the models' path: UpCatFn hands a channel slice or a copy)
                st = dskip.stride()
            add, add_bs = dskip, st[0]
        dx = torch.empty((B, C, H, W), dtype=torch.float32, device=dy.device)
        _chk(_lib().mpa_maxpool2d_bwd_add(_p(dy), ctypes.c_void_p(idx.data_ptr()), _p(add), add_bs, _p(dx), B, C, H, W,
                                         k[0], k[1], s[0], s[1], p[0], p[1], _s()), "mpa_maxpool2d_bwd_add")
        return dx, None, None, None


def pool_skip(x, kernel, stride=None, padding=(0, 0)):
    """(max_pool2d(x), skip): `skip` is x for the decoder's upconcat; its gradient is consumed in place (see PoolSkipFn)"""
    stride = kernel if stride is None else stride
    y, skip = PoolSkipFn.apply(x, tuple(kernel), tuple(stride), tuple(padding))
    skip._mpa_pool_skip = True              # upconcat(): hand the skip gradient over as a view of the concatenated one
    return y, skip


class FanoutFn(torch.autograd.Function):
    """x -> (x, x) for a tensor with two consumers (the transformer layer's residual branches, unet_cnns.py:156-158; the CNN
    family's residual stages): the two gradients are added by mpa_add here instead of by autograd's accumulation, which
    would launch an ATen kernel."""

    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)
        return x.view_as(x), x.view_as(x)

    @staticmethod
    def backward(ctx, ga, gb):
        if ga is None or gb is None:
            return ga if gb is None else gb
        ga, gb = _c(ga), _c(gb)
        out = torch.empty_like(ga)
        _chk(_lib().mpa_add(_p(ga), _p(gb), _p(out), ga.numel(), _s()), "mpa_add")
        return out


def fanout(x):
    if not (torch.is_grad_enabled() and x.requires_grad):
        return x, x
    return FanoutFn.apply(x)


class UpCatFn(torch.autograd.Function):
    """unet_up_concat_padding.forward -- unet_cnns.py:93-104.  skip_view: x2 is PoolSkipFn's skip output, whose backward
    reads its gradient in place -- the first Cs channels of dout -- so no copy of that half is made."""

    @staticmethod
    def forward(ctx, x1, x2, skip_view=False, factor=(2, 2)):
        ctx.skip_view = bool(skip_view)
        x1, x2 = _c(x1), _c(x2)
        B, C1, H1, W1 = x1.shape
        B2, Cs, Hs, Ws = x2.shape
        fh, fw = int(factor[0]), int(factor[1])
        if B != B2 or Hs < fh * H1 or Ws < fw * W1:
            raise RuntimeError(f"upconcat: skip {tuple(x2.shape)} smaller than upsampled {tuple(x1.shape)} x {factor}")
        out = torch.empty((B, Cs + C1, Hs, Ws), dtype=torch.float32, device=x1.device)
        _chk(_lib().mpa_upcat_scaled_fwd(_p(x1), _p(x2), _p(out), B, C1, H1, W1, Cs, Hs, Ws, fh, fw, _s()), "mpa_upcat_scaled_fwd")
        ctx.geom = (B, C1, H1, W1, Cs, Hs, Ws, fh, fw)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, C1, H1, W1, Cs, Hs, Ws, fh, fw = ctx.geom
        dout = _c(dout)
        dx1 = torch.empty((B, C1, H1, W1), dtype=torch.float32, device=dout.device)
        if ctx.skip_view:
            _chk(_lib().mpa_upcat_scaled_bwd(_p(dout), _p(dx1), None, B, C1, H1, W1, Cs, Hs, Ws, fh, fw, _s()),
                 "mpa_upcat_scaled_bwd")
            return dx1, dout[:, :Cs], None, None
        dskip = torch.empty((B, Cs, Hs, Ws), dtype=torch.float32, device=dout.device)
        _chk(_lib().mpa_upcat_scaled_bwd(_p(dout), _p(dx1), _p(dskip), B, C1, H1, W1, Cs, Hs, Ws, fh, fw, _s()),
             "mpa_upcat_scaled_bwd")
        return dx1, dskip, None, None


def upconcat(x1, x2, factor=(2, 2)):
    return UpCatFn.apply(x1, x2, bool(getattr(x2, "_mpa_pool_skip", False)), tuple(factor))


# --------------------------------------------------------------------------- pointwise
class ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act, slope):
        x = _c(x)
        y = torch.empty_like(x)
        _chk(_lib().mpa_act_fwd(_p(x), _p(y), x.numel(), act, float(slope), _s()), "mpa_act_fwd")
        ctx.act, ctx.slope = act, float(slope)
        ctx.save_for_backward(y)          # sign(y) == sign(x) for ReLU/LeakyReLU; sigmoid' uses y
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        _chk(_lib().mpa_act_bwd(_p(dy), _p(y), _p(dx), dy.numel(), ctx.act, ctx.slope, _s()), "mpa_act_bwd")
        return dx, None, None


def activation(x, act, slope=0.0):
    return ActFn.apply(x, act, slope)


class DropoutFn(torch.autograd.Function):
    """nn.Dropout: Bernoulli keep mask scaled by 1/(1-p); the mask is regenerated from (seed, base + offset) in backward."""

    @staticmethod
    def forward(ctx, x, p):
        x = _c(x)
        y = torch.empty_like(x)
        ctx.p, ctx.state, ctx.offset = float(p), _rng_state(x.device), _Rng.local
        _Rng.local += x.numel()
        _chk(_lib().mpa_dropout(_p(x), _p(y), x.numel(), ctx.p, _p(ctx.state), ctx.offset, _s()), "mpa_dropout")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        # same (seed, base + offset) as the forward: the base only moves in rng_advance(), after the step's backward
        _chk(_lib().mpa_dropout(_p(dy), _p(dx), dy.numel(), ctx.p, _p(ctx.state), ctx.offset, _s()), "mpa_dropout")
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    if p >= 1.0:
        raise RuntimeError("dropout p must be < 1")
    return DropoutFn.apply(x, p)


class PoolRowsDropAddFn(torch.autograd.Function):
    """MaxPool2d((kh,1), stride 1, padding (kh//2,0)) -> Dropout(p) [-> + residual] in one pass, kh = 3 or 13: the tail of
    the CNN families' prefilter stages (basic_cnns.py:374-377, with the residual add of deep_cnn_segm_sigmoid.forward,
    :414-418) and of every model's head stage conv2 (basic_cnns.py:380-385, unet_cnns.py:538-543).  Same masks and the same
    position in the dropout stream as max_pool2d + dropout + add."""

    @staticmethod
    def forward(ctx, h, residual, kh, p, producer_slope=None):
        # producer_slope: h = ReLU / LeakyReLU(conv) from conv2d(..., act_bwd_by_consumer=True): this node's backward also applies
        # that activation's (d/dh is multiplied by the slope where h <= 0 -- the sign is kept with the argmax row)
        ctx.producer_slope = producer_slope
        h = _c(h)
        B, C, H, W = h.shape
        if residual is not None:
            residual = _c(residual)
            if residual.shape != h.shape:
                raise RuntimeError(f"poolrows_dropout_add: shape mismatch {tuple(h.shape)} vs {tuple(residual.shape)}")
        out = torch.empty_like(h)
        need = ctx.needs_input_grad[0]
        which = torch.empty(h.shape, dtype=torch.int8, device=h.device) if need else None
        ctx.p, ctx.kh = float(p), int(kh)
        ctx.state, ctx.offset = (_rng_state(h.device), _Rng.local) if ctx.p > 0.0 else (None, 0)
        if ctx.p > 0.0:
            _Rng.local += h.numel()
        _chk(_lib().mpa_poolrows_dropout_add_fwd(_p(h), _p(residual) if residual is not None else None, _p(out),
                                                ctypes.c_void_p(which.data_ptr()) if need else None, B * C, H, W, ctx.kh,
                                                ctx.p, _p(ctx.state) if ctx.state is not None else None, ctx.offset,
                                                _s()), "mpa_poolrows_dropout_add_fwd")
        ctx.has_res = residual is not None
        ctx.save_for_backward(which)
        return out

    @staticmethod
    def backward(ctx, dout):
        (which,) = ctx.saved_tensors
        dout = _c(dout)
        dh = None
        if ctx.needs_input_grad[0]:
            B, C, H, W = dout.shape
            dh = torch.empty_like(dout)
            slope = 1.0 if ctx.producer_slope is None else float(ctx.producer_slope)
            _chk(_lib().mpa_poolrows_dropout_act_bwd(_p(dout), ctypes.c_void_p(which.data_ptr()), _p(dh), B * C, H, W, ctx.kh,
                                                    ctx.p, _p(ctx.state) if ctx.state is not None else None, ctx.offset,
                                                    slope, _s()), "mpa_poolrows_dropout_act_bwd")
        return dh, (dout if ctx.has_res and ctx.needs_input_grad[1] else None), None, None, None


POOLROWS_KH = (3, 13)      # window heights the fused kernels are built for


def poolrows_dropout_add(h, residual, kh, p, training, producer_slope=None):
    """dropout(max_pool2d(h, (kh,1), (1,1), (kh//2,0)), p) [+ residual]"""
    p = float(p) if training else 0.0
    if p >= 1.0:
        raise RuntimeError("dropout p must be < 1")
    if h.dim() != 4:
        raise RuntimeError("poolrows_dropout_add expects (B, C, H, W)")
    if kh not in POOLROWS_KH:
        raise RuntimeError(f"poolrows_dropout_add is built for window heights {POOLROWS_KH}, got {kh}")
    return PoolRowsDropAddFn.apply(h, residual, kh, p, producer_slope)


class AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        if a.shape != b.shape:
            raise RuntimeError(f"add: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        y = torch.empty_like(a)
        _chk(_lib().mpa_add(_p(a), _p(b), _p(y), a.numel(), _s()), "mpa_add")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return AddFn.apply(a, b)


class AddRowsFn(torch.autograd.Function):
    """x (B, ...) + pe (...): a positional table added to every sample (transformer_temporal_enc_layer, unet_cnns.py:211)."""

    @staticmethod
    def forward(ctx, x, pe):
        x, pe = _c(x, "input"), _c(pe, "positional table")
        if tuple(x.shape[1:]) != tuple(pe.shape):
            raise RuntimeError(f"add_rows: table {tuple(pe.shape)} does not match the samples {tuple(x.shape[1:])}")
        y = torch.empty_like(x)
        _chk(_lib().mpa_add_rows_bcast(_p(x), _p(pe), _p(y), x.shape[0], pe.numel(), _s()), "mpa_add_rows_bcast")
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dpe = None
        if ctx.needs_input_grad[1]:
            B, n = dy.shape[0], dy.numel() // dy.shape[0]
            dpe = torch.empty(dy.shape[1:], dtype=torch.float32, device=dy.device)
            _chk(_lib().mpa_colsum(_p(dy), _p(dpe), B, n, 0, _s()), "mpa_colsum")
        return dy, dpe


def add_rows(x, pe):
    return AddRowsFn.apply(x, pe)


class LogSoftmaxCatFn(torch.autograd.Function):
    """nn.LogSoftmax(dim=1)(torch.cat([a, b], dim=3)) -- b may be None (basic_cnns.py:254-255, 331-338)."""

    @staticmethod
    def forward(ctx, a, b):
        a = _c(a, "log-softmax input")
        b = _c(b, "log-softmax input") if b is not None else None
        B, C, R, Wa = a.shape
        Wb = 0
        if b is not None:
            if b.shape[:3] != a.shape[:3]:
                raise RuntimeError(f"logsoftmax_cat: shapes {tuple(a.shape)} and {tuple(b.shape)} differ outside the last axis")
            Wb = b.shape[3]
        y = torch.empty((B, C, R, Wa + Wb), dtype=torch.float32, device=a.device)
        _chk(_lib().mpa_logsoftmax_cat_fwd(_p(a), _p(b), _p(y), B, C, R, Wa, Wb, _s()), "mpa_logsoftmax_cat_fwd")
        ctx.save_for_backward(y)
        ctx.geom = (B, C, R, Wa, Wb)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        B, C, R, Wa, Wb = ctx.geom
        dy = _c(dy, "log-softmax grad")
        da = torch.empty((B, C, R, Wa), dtype=torch.float32, device=dy.device)
        db = torch.empty((B, C, R, Wb), dtype=torch.float32, device=dy.device) if Wb else None
        _chk(_lib().mpa_logsoftmax_cat_bwd(_p(dy), _p(y), _p(da), _p(db), B, C, R, Wa, Wb, _s()), "mpa_logsoftmax_cat_bwd")
        return da, db


def logsoftmax_cat(a, b=None):
    return LogSoftmaxCatFn.apply(a, b)


class TransposeAddFn(torch.autograd.Function):
    """x (B,R,C) -> (B,C,R), optionally adding a positional table pe (C,R) laid out like the output."""

    @staticmethod
    def forward(ctx, x, pe, pe_grad_rows):
        x = _c(x)
        B, R, C = x.shape
        y = torch.empty((B, C, R), dtype=torch.float32, device=x.device)
        pe_c = _c(pe) if pe is not None else None
        _chk(_lib().mpa_transpose_add(_p(x), _p(pe_c), _p(y), B, R, C, 1 if pe is not None else 0, _s()),
             "mpa_transpose_add")
        ctx.geom = (B, R, C)
        ctx.pe_rows = pe_grad_rows
        return y

    @staticmethod
    def backward(ctx, dy):
        B, R, C = ctx.geom
        dy = _c(dy)
        dx = torch.empty((B, R, C), dtype=torch.float32, device=dy.device)
        lib = _lib()
        _chk(lib.mpa_transpose_add(_p(dy), None, _p(dx), B, C, R, 0, _s()), "mpa_transpose_add")
        dpe = None
        if ctx.needs_input_grad[1]:
            # learnable table (max_len, E): rows [:C] receive sum_b dy[b]
            dpe = torch.zeros((ctx.pe_rows, R), dtype=torch.float32, device=dy.device)
            _chk(lib.mpa_colsum(_p(dy), _p(dpe), B, C * R, 0, _s()), "mpa_colsum")
        return dx, dpe, None


def transpose_last2(x, pe=None):
    """(B,R,C)->(B,C,R) (+pe[:C])."""
    rows = pe.shape[0] if pe is not None else 0
    pe_slice = pe[: x.shape[2]] if pe is not None else None
    if pe is not None and pe.requires_grad:
        # keep the full parameter in the graph: slicing is done by pointer (rows [:C] are a contiguous prefix)
        return TransposeAddFn.apply(x, pe, rows)
    return TransposeAddFn.apply(x, pe_slice, rows)


# --------------------------------------------------------------------------- linear / attention
class _GemmKey:
    """what the kernel probe sees of a GEMM launch (the conv launches pass their mpa_conv_desc)"""

    def __init__(self, *k):
        self._k = k

    def key(self):
        return self._k


# products the split-bf16 GEMM wins on (scratch/gemm_bfx_time.py): 128-wide tiles filled in both directions and enough k
# steps to amortise the on-the-fly split; the rest stays on the exact kernel
GEMM_BFX_MIN_WORK = 1 << 30


def _gemm(A, lda_m, lda_k, Bm, ldb_k, ldb_n, bias, C, ldc, M, N, K, accumulate=0, act=ACT_NONE):
    lib = _lib()
    if (_Precision.conv == "bf16x3" and M * N * K >= GEMM_BFX_MIN_WORK and min(M, N) >= 128 and K >= 128 and
            lib.mpa_gemm_bf16x3_supported(A, lda_m, lda_k, Bm, ldb_k, ldb_n, M, N, K)):
        call = lambda: lib.mpa_gemm_bf16x3(A, lda_m, lda_k, Bm, ldb_k, ldb_n, bias, C, ldc, M, N, K, accumulate, act, _s())
    else:
        call = lambda: lib.mpa_gemm(A, lda_m, lda_k, Bm, ldb_k, ldb_n, bias, C, ldc, M, N, K, accumulate, act, _s())
    if _Probe.match is not None:
        _chk(_probed("gemm", _GemmKey(M, N, K, int(lda_k == 1), int(ldb_k == 1), accumulate, act), call), "mpa_gemm")
    else:
        _chk(call(), "mpa_gemm")


def _ptr_off(t, off_floats=0):
    return ctypes.c_void_p(t.data_ptr() + 4 * off_floats)


def _gemm_batched(As, lda_m, lda_k, Bs, ldb_k, ldb_n, biases, Cs, ldc, M, N, K, shared_c=0, act=ACT_NONE):
    """up to 4 products of one shape in one launch; As/Bs/biases/Cs: lists of raw device addresses (ints)"""
    n = len(As)
    arr = lambda vals: (ctypes.c_void_p * n)(*vals)
    a, b, c = arr(As), arr(Bs), arr(Cs)
    bi = arr(biases) if biases is not None else None
    call = lambda: _lib().mpa_gemm_batched(n, a, lda_m, lda_k, b, ldb_k, ldb_n, bi, c, ldc, M, N, K, shared_c, act, _s())
    if _Probe.match is not None:
        _chk(_probed("gemm", _GemmKey(M, N, K, int(lda_k == 1), int(ldb_k == 1), -n if shared_c else n, act), call),
             "mpa_gemm_batched")
    else:
        _chk(call(), "mpa_gemm_batched")


class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b) over the last dim -- nn.Linear (unet_cnns.py:131-141)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act):
        x, weight, bias = _c(x), _c(weight), _c(bias)
        N, K = weight.shape
        if x.shape[-1] != K:
            raise RuntimeError(f"linear: input features {x.shape[-1]} != weight in_features {K}")
        rows = x.numel() // K
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        _gemm(_p(x), K, 1, _p(weight), 1, K, _p(bias), _p(y), N, rows, N, K, 0, act)
        ctx.act, ctx.has_bias = act, bias is not None
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _c(dy)
        N, K = weight.shape
        rows = x.numel() // K
        lib = _lib()
        if ctx.act != ACT_NONE:
            g = torch.empty_like(dy)
            _chk(lib.mpa_act_bwd(_p(dy), _p(y), _p(g), dy.numel(), ctx.act, 0.0, _s()), "mpa_act_bwd")
            dy = g
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _gemm(_p(dy), N, 1, _p(weight), K, 1, None, _p(dx), K, rows, K, N)
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(weight)
            _gemm(_p(dy), 1, N, _p(x), K, 1, None, _p(dw), K, N, K, rows)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(N, dtype=torch.float32, device=x.device)
            _chk(lib.mpa_colsum(_p(dy), _p(db), rows, N, 0, _s()), "mpa_colsum")
        return dx, dw, db, None


def linear(x, weight, bias=None, act=ACT_NONE):
    return LinearFn.apply(x, weight, bias, act)


class MlpReluFn(torch.autograd.Function):
    """y = relu(x W0^T + b0) W1^T + b1 -- the transformer layer's `mlp` (unet_cnns.py:137-141,176) as one autograd node, so that
    the ReLU's backward pass is folded into the GEMM that produces its input: dh = (dy W1) * (h > 0) is one launch
    (`mpa_gemm_masked`) instead of a product and a masking pass over the (rows, mlp_dim) tensor.  Same kernels and the same values
    as linear(linear(x, W0, b0, ACT_RELU), W1, b1) otherwise."""

    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1):
        x, w0, b0, w1, b1 = _c(x), _c(w0), _c(b0), _c(w1), _c(b1)
        H, K = w0.shape
        N = w1.shape[0]
        if x.shape[-1] != K or w1.shape[1] != H:
            raise RuntimeError(f"mlp_relu: shapes {tuple(x.shape)}, {tuple(w0.shape)}, {tuple(w1.shape)} do not chain")
        rows = x.numel() // K
        h = torch.empty(x.shape[:-1] + (H,), dtype=torch.float32, device=x.device)
        _gemm(_p(x), K, 1, _p(w0), 1, K, _p(b0), _p(h), H, rows, H, K, 0, ACT_RELU)
        y = torch.empty(x.shape[:-1] + (N,), dtype=torch.float32, device=x.device)
        _gemm(_p(h), H, 1, _p(w1), 1, H, _p(b1), _p(y), N, rows, N, H)
        ctx.has_b0, ctx.has_b1 = b0 is not None, b1 is not None
        ctx.save_for_backward(x, w0, w1, h)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w0, w1, h = ctx.saved_tensors
        dy = _c(dy)
        H, K = w0.shape
        N = w1.shape[0]
        rows = x.numel() // K
        lib = _lib()
        dx = dw0 = db0 = dw1 = db1 = None
        if ctx.needs_input_grad[3]:
            dw1 = torch.empty_like(w1)
            _gemm(_p(dy), 1, N, _p(h), H, 1, None, _p(dw1), H, N, H, rows)
        if ctx.has_b1 and ctx.needs_input_grad[4]:
            db1 = torch.empty(N, dtype=torch.float32, device=x.device)
            _chk(lib.mpa_colsum(_p(dy), _p(db1), rows, N, 0, _s()), "mpa_colsum")
        if any(ctx.needs_input_grad[:3]):
            dh = torch.empty_like(h)
            if _Precision.conv == "bf16x3" or _Probe.match is not None:      # the mode's own GEMMs / the probe's launch keys
                _gemm(_p(dy), N, 1, _p(w1), H, 1, None, _p(dh), H, rows, H, N)
                _chk(lib.mpa_act_bwd(_p(dh), _p(h), _p(dh), dh.numel(), ACT_RELU, 0.0, _s()), "mpa_act_bwd")
            else:
                _chk(lib.mpa_gemm_masked(_p(dy), N, 1, _p(w1), H, 1, _p(h), _p(dh), rows, H, N, _s()), "mpa_gemm_masked")
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _gemm(_p(dh), H, 1, _p(w0), K, 1, None, _p(dx), K, rows, K, H)
            if ctx.needs_input_grad[1]:
                dw0 = torch.empty_like(w0)
                _gemm(_p(dh), 1, H, _p(x), K, 1, None, _p(dw0), K, H, K, rows)
            if ctx.has_b0 and ctx.needs_input_grad[2]:
                db0 = torch.empty(H, dtype=torch.float32, device=x.device)
                _chk(lib.mpa_colsum(_p(dh), _p(db0), rows, H, 0, _s()), "mpa_colsum")
        return dx, dw0, db0, dw1, db1


def mlp_relu(x, w0, b0, w1, b1):
    return MlpReluFn.apply(x, w0, b0, w1, b1)


class QKVLinearFn(torch.autograd.Function):
    """q_linear / k_linear / v_linear of transformer_enc_layer (bias-free, unet_cnns.py:131-133,153) applied to the same
    tensor: three products in one launch, and in backward one launch each for the three weight gradients and for the
    input gradient (the sum of the three products)."""

    @staticmethod
    def forward(ctx, t, wq, wk, wv):
        t, wq, wk, wv = _c(t), _c(wq), _c(wk), _c(wv)
        N, K = wq.shape
        if t.shape[-1] != K or wk.shape != wq.shape or wv.shape != wq.shape:
            raise RuntimeError(f"qkv linear: input features {t.shape[-1]}, weights {tuple(wq.shape)}")
        rows = t.numel() // K
        outs = [torch.empty(t.shape[:-1] + (N,), dtype=torch.float32, device=t.device) for _ in range(3)]
        _gemm_batched([t.data_ptr()] * 3, K, 1, [w.data_ptr() for w in (wq, wk, wv)], 1, K, None,
                      [o.data_ptr() for o in outs], N, rows, N, K)
        ctx.save_for_backward(t, wq, wk, wv)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        t, wq, wk, wv = ctx.saved_tensors
        gs = [_c(g) for g in (dq, dk, dv)]
        N, K = wq.shape
        rows = t.numel() // K
        dt = torch.empty_like(t)
        _gemm_batched([g.data_ptr() for g in gs], N, 1, [w.data_ptr() for w in (wq, wk, wv)], K, 1, None,
                      [dt.data_ptr()] * 3, K, rows, K, N, shared_c=1)
        dws = [torch.empty_like(wq) for _ in range(3)]
        _gemm_batched([g.data_ptr() for g in gs], 1, N, [t.data_ptr()] * 3, K, 1, None, [d.data_ptr() for d in dws], K,
                      N, K, rows)
        return dt, dws[0], dws[1], dws[2]


def qkv_linear(t, wq, wk, wv):
    return QKVLinearFn.apply(t, wq, wk, wv)


class InProjFn(torch.autograd.Function):
    """nn.MultiheadAttention's packed in-projection: q' = q Wq^T + bq etc. with W = in_proj_weight (3E,E); the three
    products (and each kind of gradient) run as one batched launch."""

    @staticmethod
    def forward(ctx, q, k, v, w, b):
        q, k, v, w, b = _c(q), _c(k), _c(v), _c(w), _c(b)
        E = q.shape[-1]
        rows = q.numel() // E
        outs = [torch.empty_like(t) for t in (q, k, v)]
        _gemm_batched([t.data_ptr() for t in (q, k, v)], E, 1, [w.data_ptr() + 4 * i * E * E for i in range(3)], 1, E,
                      [b.data_ptr() + 4 * i * E for i in range(3)], [o.data_ptr() for o in outs], E, rows, E, E)
        ctx.save_for_backward(q, k, v, w)
        return tuple(outs)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        q, k, v, w = ctx.saved_tensors
        E = q.shape[-1]
        rows = q.numel() // E
        lib = _lib()
        gs = [_c(g) for g in (dq, dk, dv)]
        dw = torch.empty_like(w)
        db = torch.empty(3 * E, dtype=torch.float32, device=q.device)
        dins = [torch.empty_like(t) for t in (q, k, v)]
        wptr = [w.data_ptr() + 4 * i * E * E for i in range(3)]
        _gemm_batched([g.data_ptr() for g in gs], E, 1, wptr, E, 1, None, [d.data_ptr() for d in dins], E, rows, E, E)
        _gemm_batched([g.data_ptr() for g in gs], 1, E, [t.data_ptr() for t in (q, k, v)], E, 1, None,
                      [dw.data_ptr() + 4 * i * E * E for i in range(3)], E, E, E, rows)
        for i, g in enumerate(gs):
            _chk(lib.mpa_colsum(_p(g), _ptr_off(db, i * E), rows, E, 0, _s()), "mpa_colsum")
        return dins[0], dins[1], dins[2], dw, db


class AttnBatchAxisFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(d)) v over the batch axis for every position and head (Appendix C.1).  With
    `set_data_parallel_exactness(gather_attention=True)` on a data-parallel rank the keys and values of all ranks are
    gathered first (rank-major = the order of the global batch), and the backward pass sums the key / value gradients over
    the ranks and keeps this rank's slice."""

    @staticmethod
    def forward(ctx, q, k, v, heads):
        q, k, v = _c(q), _c(k), _c(v)
        B, S, E = q.shape
        world = _dp_world() if _Sync.attn else 1
        if world > 1:
            import torch.distributed as dist
            k_all = torch.empty((world * B, S, E), dtype=torch.float32, device=q.device)
            v_all = torch.empty_like(k_all)
            dist.all_gather(list(k_all.chunk(world)), k, group=_Sync.group)      # rank-major = global batch order
            dist.all_gather(list(v_all.chunk(world)), v, group=_Sync.group)
            k, v = k_all, v_all
        o = torch.empty_like(q)
        lse = torch.empty((S, heads, B), dtype=torch.float32, device=q.device)
        _chk(_lib().mpa_attn_batchaxis_fwd_kv(_p(q), _p(k), _p(v), _p(o), _p(lse), B, k.shape[0], S, E, heads, _s()),
             "mpa_attn_batchaxis_fwd_kv")
        ctx.heads, ctx.world = heads, world
        ctx.save_for_backward(q, k, v, o, lse)
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        do = _c(do)
        B, S, E = q.shape
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(k)
        _chk(_lib().mpa_attn_batchaxis_bwd_kv(_p(q), _p(k), _p(v), _p(o), _p(lse), _p(do), _p(dq), _p(dk), _p(dv), B,
                                             k.shape[0], S, E, ctx.heads, _s()), "mpa_attn_batchaxis_bwd_kv")
        if ctx.world > 1:
            import torch.distributed as dist
            rank = dist.get_rank(_Sync.group)
            dist.all_reduce(dk, op=dist.ReduceOp.SUM, group=_Sync.group)
            dist.all_reduce(dv, op=dist.ReduceOp.SUM, group=_Sync.group)
            dk, dv = dk[rank * B:(rank + 1) * B].contiguous(), dv[rank * B:(rank + 1) * B].contiguous()
        return dq, dk, dv, None


def mha_batchaxis(q, k, v, in_w, in_b, heads):
    qp, kp, vp = InProjFn.apply(q, k, v, in_w, in_b)
    return AttnBatchAxisFn.apply(qp, kp, vp, heads)


# --------------------------------------------------------------------------- BiLSTM layer
class BLSTMLayerFn(torch.autograd.Function):
    """One bidirectional nn.LSTM layer (batch_first) -- unet_cnns.py:232; gates i,f,g,o; h0=c0=0.
    args: x (B,T,I), then [w_ih, w_hh, b_ih, b_hh] for the forward and the reverse direction."""

    @staticmethod
    def forward(ctx, x, *params):
        x = _c(x)
        params = [_c(t) for t in params]
        B, T, I = x.shape
        H = params[1].shape[1]
        lib = _lib()
        dev = x.device
        out = torch.empty((B, T, 2 * H), dtype=torch.float32, device=dev)
        saved = []
        for d in range(2):
            w_ih, w_hh, b_ih, b_hh = params[4 * d: 4 * d + 4]
            bsum = torch.empty_like(b_ih)
            _chk(lib.mpa_add(_p(b_ih), _p(b_hh), _p(bsum), 4 * H, _s()), "mpa_add")
            G = torch.empty((B, T, 4 * H), dtype=torch.float32, device=dev)
            _gemm(_p(x), I, 1, _p(w_ih), 1, I, _p(bsum), _p(G), 4 * H, B * T, 4 * H, I)
            c_all = torch.empty((T, B, H), dtype=torch.float32, device=dev)
            acts = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
            order = list(range(T)) if d == 0 else list(range(T - 1, -1, -1))
            prev = None
            for t in order:
                if prev is not None:
                    # gates_t += h_prev W_hh^T ; h_prev = out[:, prev, d*H:(d+1)*H]
                    _gemm(_ptr_off(out, prev * 2 * H + d * H), T * 2 * H, 1, _p(w_hh), 1, H, None,
                          _ptr_off(G, t * 4 * H), T * 4 * H, B, 4 * H, H, accumulate=1)
                _chk(lib.mpa_lstm_cell_fwd(_ptr_off(G, t * 4 * H), T * 4 * H,
                                           _ptr_off(c_all, prev * B * H) if prev is not None else None,
                                           _ptr_off(c_all, t * B * H), _ptr_off(out, t * 2 * H + d * H), T * 2 * H,
                                           _ptr_off(acts, t * B * 4 * H), B, H, _s()), "mpa_lstm_cell_fwd")
                prev = t
            saved += [c_all, acts]
        ctx.geom = (B, T, I, H)
        ctx.save_for_backward(x, out, *params, *saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        B, T, I, H = ctx.geom
        tens = ctx.saved_tensors
        x, out = tens[0], tens[1]
        params, saved = tens[2:10], tens[10:]
        dout = _c(dout)
        lib = _lib()
        dev = x.device
        dx = torch.empty_like(x)
        grads = []
        for d in range(2):
            w_ih, w_hh, b_ih, b_hh = params[4 * d: 4 * d + 4]
            c_all, acts = saved[2 * d: 2 * d + 2]
            dG = torch.empty((B, T, 4 * H), dtype=torch.float32, device=dev)
            dh_rec = torch.empty((B, H), dtype=torch.float32, device=dev)
            dc = [torch.empty((B, H), dtype=torch.float32, device=dev) for _ in range(2)]
            order = list(range(T)) if d == 0 else list(range(T - 1, -1, -1))
            dw_hh = torch.zeros_like(w_hh)
            have_next = False
            for pos in range(T - 1, -1, -1):
                t = order[pos]
                prev = order[pos - 1] if pos > 0 else None
                _chk(lib.mpa_lstm_cell_bwd(_ptr_off(dout, t * 2 * H + d * H), T * 2 * H,
                                           _p(dh_rec) if have_next else None,
                                           _p(dc[(pos + 1) % 2]) if have_next else None,
                                           _ptr_off(acts, t * B * 4 * H),
                                           _ptr_off(c_all, prev * B * H) if prev is not None else None,
                                           _ptr_off(c_all, t * B * H), _ptr_off(dG, t * 4 * H), T * 4 * H,
                                           _p(dc[pos % 2]), B, H, _s()), "mpa_lstm_cell_bwd")
                if prev is not None:
                    # dh_rec = dgates_t W_hh ; dW_hh += dgates_t^T h_prev
                    _gemm(_ptr_off(dG, t * 4 * H), T * 4 * H, 1, _p(w_hh), H, 1, None, _p(dh_rec), H, B, H, 4 * H)
                    _gemm(_ptr_off(dG, t * 4 * H), 1, T * 4 * H, _ptr_off(out, prev * 2 * H + d * H), T * 2 * H, 1, None,
                          _p(dw_hh), H, 4 * H, H, B, accumulate=1)
                have_next = True
            dw_ih = torch.empty_like(w_ih)
            _gemm(_p(dG), 1, 4 * H, _p(x), I, 1, None, _p(dw_ih), I, 4 * H, I, B * T)
            db = torch.empty_like(b_ih)
            _chk(lib.mpa_colsum(_p(dG), _p(db), B * T, 4 * H, 0, _s()), "mpa_colsum")
            _gemm(_p(dG), 4 * H, 1, _p(w_ih), I, 1, None, _p(dx), I, B * T, I, 4 * H, accumulate=d)
            grads += [dw_ih, dw_hh, db, db]
        return (dx if ctx.needs_input_grad[0] else None, *grads)


def blstm_layer(x, params):
    return BLSTMLayerFn.apply(x, *params)


# --------------------------------------------------------------------------- losses
class BCELossFn(torch.autograd.Function):
    """torch.nn.BCELoss(reduction='mean') -- exp126a_musicnet_cnn_basic.py:87."""

    @staticmethod
    def forward(ctx, p, y):
        p, y = _c(p, "prediction"), _c(y, "target")
        if p.shape != y.shape:
            raise ValueError(f"Using a target size ({tuple(y.shape)}) that is different to the input size "
                             f"({tuple(p.shape)}) is deprecated. Please ensure they have the same size.")
        loss = torch.empty((), dtype=torch.float32, device=p.device)       # (zeroed by the call: no ATen fill kernel)
        _chk(_lib().mpa_bce_fwd(_p(p), _p(y), _p(loss), p.numel(), _s()), "mpa_bce_fwd")
        ctx.save_for_backward(p, y)
        return loss

    @staticmethod
    def backward(ctx, g):
        p, y = ctx.saved_tensors
        dp = torch.empty_like(p)
        _chk(_lib().mpa_bce_bwd(_p(p), _p(y), _p(dp), p.numel(), _p(_c(g.reshape(1))), _s()), "mpa_bce_bwd")
        return dp, None


class CrossEntropyFn(torch.autograd.Function):
    """torch.nn.CrossEntropyLoss()(logits (B,K,...1), target (B,...1)) * scale -- exp195f...py:333."""

    @staticmethod
    def forward(ctx, logits, target, scale):
        logits = _c(logits)
        B, K = logits.shape[0], logits.shape[1]
        if logits.numel() != B * K:
            raise RuntimeError("cross_entropy: only (B,K,1,1) logits are supported")
        target = target.reshape(B).to(torch.int64).contiguous()
        loss = torch.empty((), dtype=torch.float32, device=logits.device)  # (zeroed by the call)
        dl = torch.empty_like(logits)
        _chk(_lib().mpa_ce_fwd_bwd(_p(logits), ctypes.c_void_p(target.data_ptr()), _p(loss), _p(dl), B, K, float(scale),
                                  _s()), "mpa_ce_fwd_bwd")
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors      # already scaled by `scale`; an upstream factor other than 1 is applied here
        out = torch.empty_like(dl)
        _chk(_lib().mpa_scale_by(_p(dl), _p(_c(g.reshape(1))), _p(out), dl.numel(), _s()), "mpa_scale_by")
        return out, None, None
