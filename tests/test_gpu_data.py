"""GPU: patch extraction + augmentation (mpa_context_batch through the C ABI) against the data oracle and the
reference-made goldens."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from multipitch_architectures_amd.synth import synth_file

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
FILES = sorted(glob.glob(os.path.join(GOLDEN, "data_*.npz")))


def _draws_to_device_form(ds, n_harm, frames):
    """oracle draw records -> the (B,...) arrays mpa_context_batch takes"""
    B = len(ds)
    aug = np.array([[d["alpha"], d["beta"], d["tune2"], d["transp"]] for d in ds], dtype=np.int32)
    n1 = torch.stack([d["n1"] for d in ds]) if ds[0]["n1"] is not None else None
    n2 = torch.zeros(B, n_harm, frames)
    n3 = torch.zeros(B, n_harm, frames, 15)
    for b, d in enumerate(ds):
        if d["n2"] is not None:
            n2[b] = d["n2"][:, :, 0]
        if d["n3"] is not None:
            n3[b, :, :, :d["n3"].shape[2]] = d["n3"]
    return {"aug": aug, "n1": n1, "n2": n2, "n3": n3}


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[5:-4] for f in FILES])
def test_context_batch_matches_oracle_and_reference(path):
    from multipitch_architectures_amd.data_loaders import dataset_context, dataset_context_segm
    from oracle import restate_data as RD
    g = np.load(path)
    params = json.loads(str(g["params"]))
    inputs, targets = synth_file(frames=400, n_bins_out=int(g["n_out"]), seed=77)
    ti, tt = torch.from_numpy(inputs), torch.from_numpy(targets)
    cls = dataset_context_segm if "seglength" in params else dataset_context
    ds = cls(inputs, targets, dict(params))
    assert len(ds) == int(g["len"])
    items = [(int(i), int(s)) for i, s in g["items"]]
    want, draws = [], []
    for index, seed in items:
        torch.manual_seed(seed)
        X, y, d = RD.context_patch(ti, tt, params, index)
        want.append((X, y)); draws.append(d)
    Xg, yg = ds.batch([i for i, _ in items], draws=_draws_to_device_form(draws, 6, RD.n_frames(params)))
    Xg, yg = Xg.cpu(), yg.cpu()
    for k, (X, y) in enumerate(want):
        # logf on the GPU vs torch.log / np.log on the CPU: 1-2 ulp
        torch.testing.assert_close(Xg[k], X, rtol=5e-7, atol=2e-9)
        assert torch.equal(yg[k], y)
        np.testing.assert_allclose(Xg[k].numpy().ravel()[::11], g[f"{k}.xs"], rtol=5e-7, atol=2e-9)   # the reference itself
        np.testing.assert_array_equal(yg[k].numpy(), g[f"{k}.y"])


def test_context_generator_statistics_and_determinism():
    from multipitch_architectures_amd.data_loaders import dataset_context
    inputs, targets = synth_file(frames=400, seed=5)
    params = {"context": 75, "stride": 1, "compression": None, "aug:noisestd": 0.5, "aug:tuning": True,
              "aug:transpsemitones": 5, "aug:randomeq": 20}
    ds = dataset_context(np.zeros_like(inputs), targets, params, seed=3)
    idx = np.arange(64)
    aug = ds.draw(64)
    assert aug[:, 0].min() >= 1 and aug[:, 0].max() <= 20 and aug[:, 1].min() >= 0 and aug[:, 1].max() < 216
    assert set(np.unique(aug[:, 2])) <= {-2, -1, 0, 1, 2} and np.abs(aug[:, 3]).max() <= 5
    X, y = ds.batch(idx, draws={"aug": np.zeros((64, 4), np.int32)})
    v = X.flatten().double()
    # zero input + N(0, 0.5) noise, abs: half-normal -> mean = sigma*sqrt(2/pi), second moment = sigma^2
    assert abs(v.mean().item() - 0.5 * (2 / np.pi) ** 0.5) < 2e-3
    assert abs((v * v).mean().item() - 0.25) < 2e-3
    ds2 = dataset_context(np.zeros_like(inputs), targets, params, seed=3)
    X2, _ = ds2.batch(idx, draws={"aug": np.zeros((64, 4), np.int32)})
    assert torch.equal(X, X2)                               # same seed, same call number -> same noise
    X3, _ = ds2.batch(idx, draws={"aug": np.zeros((64, 4), np.int32)})
    assert not torch.equal(X, X3)                           # next call -> fresh noise
    # refilled edges are |N(0,1e-4)|: transposition +2 semitones -> bins 0..5
    a = np.zeros((64, 4), np.int32); a[:, 3] = 2
    X4, y4 = ds.batch(idx, draws={"aug": a})
    e = X4[:, :, :, :6].flatten().double()
    assert 0 < e.mean().item() < 3e-4 and abs(e.mean().item() - 1e-4 * (2 / np.pi) ** 0.5) < 5e-6
    assert float(y4[:, :, :, :2].abs().max()) == 0


def test_context_loader_covers_every_patch_once_and_shards():
    from multipitch_architectures_amd.data_loaders import ContextLoader, dataset_context
    params = {"context": 75, "stride": 50, "compression": 10}
    files = [synth_file(frames=f, seed=s) for f, s in ((400, 1), (333, 2), (180, 3))]
    dss = [dataset_context(i, t, params) for i, t in files]
    total = sum(len(d) for d in dss)
    full = [dss[k].batch(np.arange(len(dss[k])))[0] for k in range(3)]
    ref = torch.cat(full)
    got = torch.cat([X for X, _ in ContextLoader(dss, batch_size=4, shuffle=False)])
    assert got.shape[0] == total and torch.equal(got, ref)
    # shuffled: a permutation of the same patches
    sh = torch.cat([X for X, _ in ContextLoader(dss, batch_size=5, shuffle=True, seed=9)])
    key = lambda t: sorted(t.flatten(1).sum(1).tolist())
    assert sh.shape == ref.shape and key(sh) == key(ref)
    # two ranks split every batch
    r0 = [X for X, _ in ContextLoader(dss, batch_size=4, shuffle=False, rank=0, world=2)]
    r1 = [X for X, _ in ContextLoader(dss, batch_size=4, shuffle=False, rank=1, world=2)]
    assert sum(x.shape[0] for x in r0) + sum(x.shape[0] for x in r1) == total
    assert torch.equal(torch.cat([torch.cat([a, b]) for a, b in zip(r0, r1)] + r0[len(r1):]), ref)


def test_context_rejects_bad_requests():
    from multipitch_architectures_amd.data_loaders import dataset_context
    inputs, targets = synth_file(frames=200, seed=5)
    ds = dataset_context(inputs, targets, {"context": 75, "stride": 10, "compression": 10})
    with pytest.raises(IndexError):
        ds.batch([len(ds) + 3])
    scaled = dataset_context(inputs, targets, {"context": 75, "stride": 10, "compression": 10, "aug:scalingfactor": 1.2})
    with pytest.raises(AssertionError, match="Scaling not implemented for dataset_context"):      # upstream's own (:77-78)
        scaled[0]
    X, y = ds[0]
    assert tuple(X.shape) == (6, 75, 216) and tuple(y.shape) == (1, 1, 72)


VARIANT_FILES = sorted(glob.glob(os.path.join(GOLDEN, "datax*.npz")))


@pytest.mark.parametrize("path", VARIANT_FILES, ids=[os.path.basename(f)[5:-4] for f in VARIANT_FILES])
def test_slicing_dataset_variants(path):
    """dataset_context_segm_pitch / _widetarget / dataset_context_measuresegm (hcqt_datasets.py:292-436) and the target smoothing of
    dataset_context_segm ('aug:smooth_len', :190-194): no random decisions, so the HIP classes are compared with the reference
    classes' own output (oracle/make_goldens_data.py --variants): len(), X (strided sample + sums, logf vs np.log: 1-2 ulp), y"""
    from multipitch_architectures_amd import data_loaders as DL
    g = np.load(path)
    params, cls_name = json.loads(str(g["params"])), str(g["cls"])
    inputs, targets = synth_file(frames=int(g["frames"]), n_bins_out=int(g["n_out"]), seed=78)
    args = (inputs, targets) + ((g["measures"],) if cls_name == "dataset_context_measuresegm" else ())
    ds = getattr(DL, cls_name)(*args, dict(params))
    assert len(ds) == int(g["len"])
    scaling = "aug:scalingfactor" in params       # time scaling (:211-225): the drawn factor is stored with the item
    for k, index in enumerate(g["indices"]):
        if scaling:
            Xb, yb = ds.batch([int(index)], draws={"aug": np.zeros((1, 4), np.int32), "scale": float(g[f"{k}.scale"])})
            X, y = Xb[0], yb[0]
            assert X.shape[1] == int(g[f"{k}.new_len"]) + 2 * (params["context"] // 2)
        else:
            X, y = ds[int(index)]
        X, y = X.cpu().numpy(), y.cpu().numpy()
        assert list(X.shape) == list(g[f"{k}.shape"]) and y.shape == g[f"{k}.y"].shape
        np.testing.assert_allclose(X.ravel()[::11], g[f"{k}.xs"], rtol=5e-7, atol=2e-9)
        s = g[f"{k}.stats"]
        np.testing.assert_allclose([X.astype(np.float64).sum(), np.abs(X).astype(np.float64).sum(), X.max()], s, rtol=1e-6)
        if "aug:smooth_len" in params:       # smoothed targets: float64 convolution cast to float32 on both sides
            np.testing.assert_allclose(y, g[f"{k}.y"], rtol=1e-6, atol=1e-7)
        else:
            np.testing.assert_array_equal(y, g[f"{k}.y"])
    if scaling:                                             # own random factor per item: lengths within the stated range, one per call
        lo, hi = params["seglength"] / params["aug:scalingfactor"], params["seglength"] * (2 - 1 / params["aug:scalingfactor"])
        lens = {ds[0][0].shape[1] - 2 * (params["context"] // 2) for _ in range(12)}
        assert len(lens) > 1 and all(int(lo) <= n <= int(hi) for n in lens)
        with pytest.raises(RuntimeError):
            ds.batch([0, 1])
    elif cls_name != "dataset_context_measuresegm":         # batches of equal-length segments stack
        Xb, yb = ds.batch([int(i) for i in g["indices"]])
        assert Xb.shape[0] == len(g["indices"]) and torch.equal(Xb[1], ds[int(g["indices"][1])][0])
    else:
        with pytest.raises(RuntimeError):
            ds.batch([0, 1])
