"""GPU-resident replacement of ``libdl/data_loaders/hcqt_datasets.py`` ``dataset_context`` (:10-141) and
``dataset_context_segm`` (:144-289): same constructor, ``len()`` and parameter dictionary; the recording lives in HBM
and patches are cut *and* augmented by one HIP launch per batch (``mpa_context_batch``) instead of by 16 DataLoader
workers (exp180d...py:54-57).

Differences, all deliberate:
* ``__getitem__`` returns device tensors; ``batch(indices)`` is the fast path, ``ContextLoader`` the
  ``ConcatDataset`` + ``DataLoader(batch_size, shuffle)`` equivalent (with rank sharding for data-parallel runs).
* Random decisions (EQ parabola, tuning shift, transposition) are drawn on the host from a numpy PCG64 stream, noise by
  a counter-based generator in the kernel -- same distributions as the reference, not the same torch-CPU stream.  Pass
  ``draws=...`` to ``batch`` to dictate them (used by the parity tests).
* ``'aug:scalingfactor'`` (time scaling, ``dataset_context_segm`` only, :211-225) resamples every patch to its own random
  length, so its samples do not stack: they are served one per call (``ds[i]`` / ``batch([i])``; upstream a DataLoader can
  only batch them with batch_size 1 as well), the input ``int(scalefac * seglength)`` frames plus context long beside
  ``seglength`` target rows, as upstream.  ``dataset_context`` with the key raises upstream's assertion when an item is
  requested (:77-78).
Round 4: ``'aug:smooth_len'`` / ``'aug:smooth_win'`` (target smoothing at construction, :190-194) and the three plain-slicing
classes ``dataset_context_segm_pitch`` (:292-330), ``dataset_context_segm_widetarget`` (:333-377) and
``dataset_context_measuresegm`` (:380-436) on the same launch.
There is no CPU path: construction fails without the HIP library and a GPU.
"""
import ctypes

import numpy as np
import torch

from .. import _lib as L

N_BINS = 216


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _eq_min_ok(alpha, beta, n_harm):
    """vectorised form of the reference's accept test ``min(filtmat) >= 0`` (hcqt_datasets.py:84-97): the parabola's
    minimum over bins 0..215 is at the bin farthest from its centre; evaluated in float32 like the reference."""
    ok = np.ones(alpha.shape, dtype=bool)
    scale = np.float32(2e-6) * alpha.astype(np.float32)
    for h in range(n_harm):
        off = -36 if h == 0 else int(36 * np.log2(h))
        c = beta - off
        far = np.maximum(np.abs(0 - c), np.abs(N_BINS - 1 - c)).astype(np.int64)
        ok &= (np.float32(1) - scale * (far * far).astype(np.float32)) >= 0
    return ok


class dataset_context:
    """``dataset_context(inputs, targets, params)`` -- hcqt_datasets.py:30-60.

    inputs: (n_harm, T, 216) tensor/array of *uncompressed* HCQT magnitudes, targets: (T, n_out)."""

    _segm = False

    def __init__(self, inputs, targets, params, device="cuda:0", seed=0):
        L.load()                                        # fail loudly: no CPU path
        if not torch.cuda.is_available():
            raise RuntimeError("dataset_context needs a GPU: the recordings are kept resident in HBM")
        self.device = torch.device(device)
        self.inputs = torch.as_tensor(inputs).to(self.device, torch.float32).contiguous()
        self.targets = torch.as_tensor(targets).to(self.device, torch.float32).contiguous()
        if self.inputs.dim() != 3 or self.targets.dim() != 2 or self.inputs.shape[1] != self.targets.shape[0]:
            raise RuntimeError(f"expected inputs (n_harm,T,n_bins) and targets (T,n_out); got "
                               f"{tuple(self.inputs.shape)} / {tuple(self.targets.shape)}")
        self.context = params["context"]
        self.stride = params["stride"]
        self.compression = params["compression"]
        self.seglength = params["seglength"] if self._segm else 1
        self.targettype = params.get("targettype", "pitch_class")
        self.transposition = params.get("aug:transpsemitones")
        self.scalingfactor = params.get("aug:scalingfactor")
        self.randomeq = params.get("aug:randomeq")
        self.noisestd = params.get("aug:noisestd")
        self.tuning = params.get("aug:tuning")
        if params.get("aug:smooth_len", 0) > 1:
            if not self._segm:
                raise NotImplementedError("'aug:smooth_len' belongs to dataset_context_segm (hcqt_datasets.py:190-194)")
            # one-off at construction, as upstream: the targets convolved along time with a window, scaled to maximum 1
            from scipy import signal
            filt = np.expand_dims(signal.get_window(params["aug:smooth_win"], params["aug:smooth_len"] + 1)[1:], axis=1)
            sm = signal.convolve(np.asarray(torch.as_tensor(targets).cpu()), filt, mode="same")
            sm /= np.max(sm)
            self.targets = torch.from_numpy(sm).to(self.device, torch.float32).contiguous()
        if (self.randomeq or self.transposition) and self.inputs.shape[2] != N_BINS:
            raise RuntimeError("the augmentations assume 216 bins (3 per semitone), as the reference does")
        if self.transposition and self.transposition > 5:
            raise RuntimeError("'aug:transpsemitones' > 5 is not supported")
        self.rng = np.random.Generator(np.random.PCG64(seed))
        self._calls = 0
        self._seed = seed
        flags = 0
        flags |= L.CTX_EQ if self.randomeq else 0
        flags |= L.CTX_NOISE if self.noisestd else 0
        flags |= L.CTX_LOG if self.compression is not None else 0
        flags |= L.CTX_TUNE if self.tuning else 0
        flags |= L.CTX_TRANSP if self.transposition else 0
        flags |= L.CTX_SEGM_TARGETS if self._segm else 0
        self.frames = 2 * (self.context // 2) + self.seglength
        self.desc = L.ContextDesc(self.inputs.shape[0], self.inputs.shape[2], self.frames, self.targets.shape[1],
                                  self.seglength, flags, float(self.compression or 0.0), float(self.noisestd or 0.0))

    def __len__(self):
        if self._segm:                                  # :195-197
            return (self.inputs.shape[1] - self.context - self.seglength + self.stride) // self.stride
        return (self.inputs.shape[1] - self.context) // self.stride   # :62-64

    # ------------------------------------------------------------------ random decisions (host side)
    def draw(self, n):
        """(n,4) int32: alpha, beta, tune2, transp -- distributions of hcqt_datasets.py:83-97,109,127."""
        aug = np.zeros((n, 4), dtype=np.int32)
        if self.randomeq:
            todo = np.arange(n)
            while todo.size:
                a = self.rng.integers(1, self.randomeq + 1, todo.size)
                b = self.rng.integers(0, N_BINS, todo.size)
                ok = _eq_min_ok(a, b, self.inputs.shape[0])
                aug[todo[ok], 0], aug[todo[ok], 1] = a[ok], b[ok]
                todo = todo[~ok]
        if self.tuning:
            aug[:, 2] = self.rng.integers(-2, 3, n)
        if self.transposition:
            aug[:, 3] = self.rng.integers(-self.transposition, self.transposition + 1, n)
        return aug

    # ------------------------------------------------------------------ addressing (host side, bounds-checked)
    def addresses(self, indices):
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        start = idx * self.stride
        # like the reference, an index past len() is served as long as its window lies inside the recording; anything
        # else would read out of bounds on the device and is refused here
        if idx.size and (idx.min() < 0 or start.max() + self.frames > self.inputs.shape[1]):
            raise IndexError(f"patch window outside the recording (len() = {len(self)}, frames = {self.inputs.shape[1]})")
        src = self.inputs.data_ptr() + start * (self.inputs.shape[2] * 4)
        tgt = self.targets.data_ptr() + (start + self.context // 2) * (self.targets.shape[1] * 4)
        cs = np.full(idx.shape, self.inputs.shape[1] * self.inputs.shape[2], dtype=np.int64)
        return src.astype(np.uint64), cs, tgt.astype(np.uint64)

    def batch(self, indices, draws=None):
        if self.scalingfactor:
            raise AssertionError("Scaling not implemented for dataset_context!")      # hcqt_datasets.py:77-78
        return gather([(self, indices)], draws=draws)

    def __getitem__(self, index):
        X, y = self.batch([index])
        return X[0], y[0]


class dataset_context_segm(dataset_context):
    """``dataset_context_segm`` (hcqt_datasets.py:144-289): ``seglength`` target frames per patch.  With
    ``'aug:scalingfactor'`` the frames between the context halves are first resampled to ``int(scalefac * seglength)`` frames,
    ``scalefac = 1/f + 2 u (1 - 1/f)``, u uniform in [0, 1) (:212-213; ``draws["scale"]`` dictates it); one patch per call."""
    _segm = True

    def batch(self, indices, draws=None):
        if not self.scalingfactor:
            return gather([(self, indices)], draws=draws)
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        if idx.size != 1:
            raise RuntimeError("'aug:scalingfactor' serves one patch per call: their lengths differ")
        src, cs, _ = dataset_context.addresses(self, idx)          # bounds-checked window incl. context
        if draws is not None and draws.get("scale") is not None:
            scalefac = float(draws["scale"])
        else:
            scalefac = 1.0 / self.scalingfactor + 2.0 * float(self.rng.random()) * (1.0 - 1.0 / self.scalingfactor)
        new_len = int(scalefac * self.seglength)
        if new_len < 1:
            raise RuntimeError(f"'aug:scalingfactor': scaled length {new_len} < 1")
        hc = self.context // 2
        frames = new_len + 2 * hc
        scaled = torch.empty((self.inputs.shape[0], frames, self.inputs.shape[2]), dtype=torch.float32, device=self.device)
        L.check(L.load().mpa_time_scale(ctypes.c_void_p(int(src[0])), ctypes.c_int64(int(cs[0])), self.inputs.shape[0],
                                        self.inputs.shape[2], hc, self.seglength, new_len, _p(scaled),
                                        ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "mpa_time_scale")
        # the remaining stages on the scaled window: same launch as every other patch, with its own frame count
        keep = self.frames, self.desc
        self.frames = frames
        self.desc = L.ContextDesc(self.inputs.shape[0], self.inputs.shape[2], frames, self.targets.shape[1], self.seglength,
                                  self.desc.flags, float(self.compression or 0.0), float(self.noisestd or 0.0))
        self._scaled = (scaled, idx)
        try:
            X, y = gather([(self, idx)], draws=None if draws is None else {k: v for k, v in draws.items() if k != "scale"})
        finally:
            self.frames, self.desc = keep
            self._scaled = None
        X._mpa_keepalive = X._mpa_keepalive + (scaled,)
        return X, y

    _scaled = None

    def addresses(self, indices):
        src, cs, tgt = dataset_context.addresses(self, indices) if self._scaled is None else self._addresses_scaled()
        return src, cs, tgt

    def _addresses_scaled(self):
        scaled, idx = self._scaled
        start = idx * self.stride
        tgt = self.targets.data_ptr() + (start + self.context // 2) * (self.targets.shape[1] * 4)
        return (np.array([scaled.data_ptr()], dtype=np.uint64), np.array([scaled.shape[1] * scaled.shape[2]], dtype=np.int64),
                tgt.astype(np.uint64))


class dataset_context_segm_pitch(dataset_context_segm):
    """``dataset_context_segm_pitch`` (hcqt_datasets.py:292-330): segments without augmentation whose targets are columns
    24..95 of a 128-pitch roll.  The columns are cut once at construction (a resident (T,72) tensor)."""

    def __init__(self, inputs, targets, params, device="cuda:0", seed=0):
        t = torch.as_tensor(targets)
        if t.dim() != 2 or t.shape[1] < 96:
            raise RuntimeError(f"dataset_context_segm_pitch: targets must be (T, >= 96) pitch rolls, got {tuple(t.shape)}")
        keep = {k: params[k] for k in ("context", "seglength", "stride", "compression")}
        super().__init__(inputs, t[:, 24:96], keep, device=device, seed=seed)


class dataset_context_segm_widetarget(dataset_context_segm):
    """``dataset_context_segm_widetarget`` (hcqt_datasets.py:333-377): `seglength` target frames, and around their centre an
    input window of 500 frames + context (no augmentation).  Indices whose window leaves the recording raise IndexError
    (upstream a negative slice start silently yields a wrong-sized tensor)."""
    SEGL_HCQT = 500

    def __init__(self, inputs, targets, params, device="cuda:0", seed=0):
        keep = {k: params[k] for k in ("context", "seglength", "stride", "compression")}
        super().__init__(inputs, targets, keep, device=device, seed=seed)
        self.frames = 2 * (self.context // 2) + self.SEGL_HCQT
        self.desc = L.ContextDesc(self.inputs.shape[0], self.inputs.shape[2], self.frames, self.targets.shape[1],
                                  self.seglength, self.desc.flags, float(self.compression or 0.0), 0.0)

    def addresses(self, indices):
        idx = np.asarray(indices, dtype=np.int64).reshape(-1)
        index = idx * self.stride + self.context // 2
        x0 = index + self.seglength // 2 - self.SEGL_HCQT // 2 - self.context // 2
        if idx.size and (idx.min() < 0 or x0.min() < 0 or x0.max() + self.frames > self.inputs.shape[1] or
                         index.max() + self.seglength > self.targets.shape[0]):
            raise IndexError("patch window outside the recording")
        src = self.inputs.data_ptr() + x0 * (self.inputs.shape[2] * 4)
        tgt = self.targets.data_ptr() + index * (self.targets.shape[1] * 4)
        cs = np.full(idx.shape, self.inputs.shape[1] * self.inputs.shape[2], dtype=np.int64)
        return src.astype(np.uint64), cs, tgt.astype(np.uint64)


class dataset_context_measuresegm(dataset_context_segm):
    """``dataset_context_measuresegm`` (hcqt_datasets.py:380-436): segments bounded by given measure positions (frame
    indices); `seglength` and `stride` count measures.  Segments have different lengths, so items are served one by one
    (``ds[i]``; upstream a DataLoader can only batch them with batch_size 1 as well)."""

    def __init__(self, inputs, targets, measures, params, device="cuda:0", seed=0):
        keep = {k: params[k] for k in ("context", "seglength", "stride", "compression")}
        self._measure_seglength = keep["seglength"]
        super().__init__(inputs, targets, dict(keep, seglength=1), device=device, seed=seed)
        self.measures = np.asarray(torch.as_tensor(measures).cpu()).astype(np.float64).reshape(-1)

    def __len__(self):
        return (self.measures.size - self._measure_seglength - 1) // self.stride     # :416-418

    def batch(self, indices, draws=None):
        idx = np.asarray(indices).reshape(-1)
        if idx.size != 1:
            raise RuntimeError("dataset_context_measuresegm serves one segment per call: their lengths differ")
        i = int(idx[0]) * self.stride
        start, end = int(self.measures[i]), int(self.measures[i + self._measure_seglength])
        hc = self.context // 2
        if i < 0 or end <= start or start - hc < 0 or end + hc > self.inputs.shape[1]:
            raise IndexError("measure segment (with its context) outside the recording")
        self.seglength, self.frames = end - start, end - start + 2 * hc
        self.desc = L.ContextDesc(self.inputs.shape[0], self.inputs.shape[2], self.frames, self.targets.shape[1],
                                  self.seglength, self.desc.flags, float(self.compression or 0.0), 0.0)
        self._start = start
        return gather([(self, idx)])

    def addresses(self, indices):
        hc = self.context // 2
        src = np.array([self.inputs.data_ptr() + (self._start - hc) * (self.inputs.shape[2] * 4)], dtype=np.uint64)
        tgt = np.array([self.targets.data_ptr() + self._start * (self.targets.shape[1] * 4)], dtype=np.uint64)
        cs = np.array([self.inputs.shape[1] * self.inputs.shape[2]], dtype=np.int64)
        return src, cs, tgt


def gather(parts, draws=None):
    """One launch for patches taken from several resident recordings.
    parts: [(dataset, indices)], all datasets sharing the parameter dictionary (the first one's is used).
    draws: optional dict(aug=(B,4) int, n1=..., n2=..., n3=... float32 tensors or None) to dictate the randomness."""
    ds0 = parts[0][0]
    src, cs, tgt = (np.concatenate(a) for a in zip(*[ds.addresses(ix) for ds, ix in parts]))
    B = int(src.size)
    dev = ds0.device
    aug = np.asarray(draws["aug"], dtype=np.int32).reshape(B, 4) if draws is not None else ds0.draw(B)
    table = torch.from_numpy(np.concatenate([src.view(np.int64), cs, tgt.view(np.int64)])).to(dev, non_blocking=True)
    aug_d = torch.from_numpy(aug).to(dev, non_blocking=True)
    X = torch.empty((B, ds0.desc.n_harm, ds0.frames, ds0.desc.n_bins), dtype=torch.float32, device=dev)
    y = torch.empty((B, 1, ds0.seglength, ds0.desc.n_out), dtype=torch.float32, device=dev)
    n1 = n2 = n3 = None
    if draws is not None:
        n1, n2, n3 = (None if draws.get(k) is None else draws[k].to(dev, torch.float32).contiguous()
                      for k in ("n1", "n2", "n3"))
        if n1 is not None: assert tuple(n1.shape) == tuple(X.shape)
        if n2 is not None: assert n2.numel() == B * ds0.desc.n_harm * ds0.frames
        if n3 is not None: assert n3.numel() == B * ds0.desc.n_harm * ds0.frames * 15
    ds0._calls += 1
    seed = (ds0._seed * 0x9E3779B97F4A7C15 + ds0._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    off = table.data_ptr()
    rc = L.load().mpa_context_batch(ctypes.byref(ds0.desc), B, ctypes.c_void_p(off), ctypes.c_void_p(off + 8 * B),
                                    ctypes.c_void_p(off + 16 * B), _p(aug_d), _p(n1), _p(n2), _p(n3),
                                    ctypes.c_uint64(seed), _p(X), _p(y),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    L.check(rc, "mpa_context_batch")
    X._mpa_keepalive = (table, aug_d)      # the launch is asynchronous w.r.t. the host
    if ds0._segm:                          # the reference's per-item target is (1,1,seglength,n_out) there (:208)
        y = y.view(B, 1, 1, ds0.seglength, ds0.desc.n_out)
    return X, y


class ContextLoader:
    """``DataLoader(ConcatDataset(datasets), batch_size, shuffle)`` (exp180d...py:287-288) over resident recordings.
    ``rank``/``world`` shard every global batch like ``parallel.shard_range``."""

    def __init__(self, datasets, batch_size, shuffle=False, seed=0, rank=0, world=1, drop_last=False):
        self.datasets = list(datasets)
        self.batch_size, self.shuffle, self.drop_last = batch_size, shuffle, drop_last
        self.rank, self.world = rank, world
        self.rng = np.random.Generator(np.random.PCG64(seed))
        lens = np.array([len(d) for d in self.datasets], dtype=np.int64)
        self.file_of = np.repeat(np.arange(len(self.datasets)), lens)
        self.local = np.concatenate([np.arange(n) for n in lens]) if lens.size else np.zeros(0, np.int64)

    def __len__(self):
        n = self.file_of.size
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        order = self.rng.permutation(self.file_of.size) if self.shuffle else np.arange(self.file_of.size)
        for k in range(len(self)):
            sel = order[k * self.batch_size:(k + 1) * self.batch_size]
            per = -(-sel.size // self.world)
            sel = sel[self.rank * per:(self.rank + 1) * per]
            if sel.size == 0:
                continue
            parts = []
            for f in np.unique(self.file_of[sel]):           # group by recording, keep the batch order inside a group
                parts.append((self.datasets[f], self.local[sel[self.file_of[sel] == f]]))
            yield gather(parts)
