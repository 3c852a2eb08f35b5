"""CPU-only checks of the host side: the C-ABI library loads and exports every symbol include/mpa.h declares,
the drop-in import surface, constructor signatures, loud failure without a GPU, synthetic-data determinism."""
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "mpa.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mpa_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from multipitch_architectures_amd import _lib
    lib = _lib.load(build_if_missing=True)
    syms = _header_symbols()
    assert len(syms) >= 40
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/mpa.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes table and header disagree"
    assert lib.mpa_version() >= 1
    assert lib.mpa_strerror(-3) == b"unsupported configuration"


def test_c_abi_rejects_bad_arguments_without_a_gpu():
    """argument validation happens before any launch, so it is checkable on the CPU"""
    import ctypes
    from multipitch_architectures_amd import _lib
    lib = _lib.load(build_if_missing=True)
    d = _lib.ConvDesc(1, 6, 75, 216, 16, 15, 15, 1, 1, 7, 7)
    assert lib.mpa_conv2d_packed_floats(ctypes.byref(d), 0) > 0
    assert lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d)) > 0
    assert lib.mpa_conv2d_fwd(ctypes.byref(d), None, None, None, None, 0, 0.0, None) == -1
    strided = _lib.ConvDesc(1, 6, 75, 216, 16, 3, 3, 2, 2, 1, 1)
    assert lib.mpa_conv2d_packed_floats(ctypes.byref(strided), 1) == -3      # unsupported bwd-data geometry
    assert lib.mpa_gemm(None, 1, 1, None, 1, 1, None, None, 1, 4, 4, 4, 0, 0, None) == -1


def test_import_surface_matches_reference():
    from multipitch_architectures_amd import nn_models
    names = """basic_cnn basic_cnn_pool basic_cnn_segm_sigmoid basic_cnn_segm_logsoftmax basic_cnn_segm_blank_logsoftmax
    deep_cnn_segm_sigmoid single_conv double_conv unet_up_concat_padding transformer_enc_layer simple_u_net
    simple_u_net_largekernels simple_u_net_selfattn simple_u_net_doubleselfattn freq_u_net freq_u_net_bottomstack
    freq_u_net_selfattn freq_u_net_doubleselfattn simple_u_net_doubleselfattn_twolayers
    simple_u_net_doubleselfattn_alllayers simple_u_net_doubleselfattn_varlayers simple_u_net_sixselfattn
    u_net_temporal_selfattn_varlayers transformer_temporal_enc_layer simple_u_net_doubleselfattn_transenc
    blstm_temporal_enc_layer u_net_blstm_varlayers u_net_temporal_blstm_varlayers simple_u_net_doubleselfattn_polyphony
    simple_u_net_doubleselfattn_polyphony_classif simple_u_net_polyphony_classif
    simple_u_net_polyphony_classif_softmax""".split()
    assert len(names) == 32
    for n in names:
        assert hasattr(nn_models, n), n
    # the three names the reference itself cannot construct fail with the reference's exception classes (Appendix C.7)
    with pytest.raises(NameError):
        nn_models.freq_u_net()
    with pytest.raises(NameError):
        nn_models.freq_u_net_bottomstack()
    with pytest.raises(UnboundLocalError):
        nn_models.single_conv(4, 4)
    assert len(nn_models.BUILT) + len(nn_models.NOT_BUILT) == 32 and len(nn_models.NOT_BUILT) == 3


def test_constructor_signatures_verbatim():
    """keyword names, order and defaults of the reference constructors"""
    from multipitch_architectures_amd import nn_models as M
    def sig(c):
        return [(p.name, p.default) for p in list(inspect.signature(c.__init__).parameters.values())[1:]]
    assert sig(M.basic_cnn_segm_sigmoid) == [("n_chan_input", 6), ("n_chan_layers", [20, 20, 10, 1]), ("n_bins_in", 216),
                                             ("n_bins_out", 12), ("a_lrelu", 0.3), ("p_dropout", 0.2)]
    assert sig(M.deep_cnn_segm_sigmoid) == [("n_chan_input", 6), ("n_chan_layers", [20, 20, 10, 1]),
                                            ("n_prefilt_layers", 1), ("residual", False), ("n_bins_in", 216),
                                            ("n_bins_out", 12), ("a_lrelu", 0.3), ("p_dropout", 0.2)]
    assert sig(M.simple_u_net_largekernels) == [("n_chan_input", 6), ("n_chan_layers", [64, 30, 20, 10]),
                                                ("n_bins_in", 216), ("n_bins_out", 12), ("a_lrelu", 0.3),
                                                ("p_dropout", 0.2), ("scalefac", 16)]
    assert sig(M.simple_u_net_doubleselfattn)[6:] == [("convdrop", 0), ("residual", False), ("alt_order", False),
                                                      ("scalefac", 16), ("embed_dim", 32), ("num_heads", 8),
                                                      ("mlp_dim", 512), ("pos_encoding", None)]
    assert sig(M.simple_u_net_doubleselfattn_twolayers)[6:] == [("convdrop", 0), ("residual", False), ("scalefac", 16),
                                                                ("embed_dim", 32), ("num_heads", 8), ("mlp_dim", 512),
                                                                ("pos_encoding", None)]
    assert sig(M.u_net_blstm_varlayers)[6:] == [("scalefac", 8), ("embed_dim", 64), ("hidden_size", 512),
                                                ("lstm_depth", 0), ("lstm_number", 2)]
    assert sig(M.simple_u_net_polyphony_classif_softmax)[6:] == [("scalefac", 16), ("num_polyphony_steps", 24)]
    assert sig(M.double_conv) == [("in_channels", inspect._empty), ("out_channels", inspect._empty), ("mid_channels", None),
                                  ("kernel_size", (3, 3)), ("padding", (1, 1)), ("convdrop", 0), ("residual", False),
                                  ("alt_order", False)]
    assert sig(M.transformer_enc_layer) == [("embed_dim", 32), ("num_heads", 8), ("mlp_dim", 512), ("p_dropout", 0.2),
                                            ("pos_encoding", None)]
    assert sig(M.blstm_temporal_enc_layer) == [("embed_dim", 32), ("hidden_size", 512), ("num_layers", 1),
                                               ("batch_first", True), ("bidirectional", True)]


def test_parameter_counts_match_reference_logs():
    """params of the paper configurations (experiments/logs/**: 'Total params')"""
    from multipitch_architectures_amd import nn_models as M
    from multipitch_architectures_amd.configs import CONFIGS
    expect = {"CNN:XS": 48255, "DRCNN:L": 4814683, "Unet:L": 4552227, "SAUnet:L": 8115003, "BLUnet:XXL": 22376255,
              "PUnet:XL": 14597963, "SAUnet:M": 1179911 + 2 * (3 * 64 * 64 + 3 * 64 + 64 * 64 + 64)}
    for name, n in expect.items():
        cfg = CONFIGS[name]
        m = getattr(M, cfg["cls"])(**cfg["kwargs"])
        assert sum(p.numel() for p in m.parameters()) == n, name


def test_cpu_tensors_raise_not_fallback():
    from helpers import build_model
    from multipitch_architectures_amd.synth import synth_batch
    m = build_model("tiny:CNN", "cpu")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(synth_batch(1, 75)[0])
    from multipitch_architectures_amd.losses import BCELoss
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        BCELoss()(torch.rand(2, 3), torch.rand(2, 3))
    from multipitch_architectures_amd.optim import AdamW
    p = torch.nn.Parameter(torch.zeros(3))
    p.grad = torch.ones(3)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AdamW([p]).step()
    from multipitch_architectures_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.poolrows_dropout_add(torch.zeros(1, 2, 5, 4), None, 3, 0.2, True)


def test_sinusoidal_pe_not_in_state_dict_and_constructible_without_gpu():
    """Appendix C.3: pe is a plain attribute; unlike the reference the ctor must not need a GPU"""
    from multipitch_architectures_amd.nn_models import transformer_enc_layer
    layer = transformer_enc_layer(embed_dim=32, num_heads=8, mlp_dim=64, pos_encoding="sinusoidal")
    assert "pe" not in layer.state_dict() and layer.pe.shape == (600, 32)
    assert len(layer.state_dict()) == 16
    learn = transformer_enc_layer(embed_dim=32, num_heads=8, mlp_dim=64, pos_encoding="learnable")
    assert "pe" in learn.state_dict()


def test_synth_is_deterministic():
    from multipitch_architectures_amd.synth import det_fill, synth_batch
    a, ya = synth_batch(3, 80, seed=5)
    b, yb = synth_batch(3, 80, seed=5)
    assert torch.equal(a, b) and torch.equal(ya, yb)
    assert a.shape == (3, 6, 80, 216) and ya.shape == (3, 1, 6, 72) and a.min() >= 0
    sd = {"x.weight": torch.zeros(4, 3, 3, 3), "x.bias": torch.zeros(4), "bn.running_var": torch.zeros(4)}
    f1, f2 = det_fill(sd), det_fill(sd)
    assert all(torch.equal(f1[k], f2[k]) for k in sd) and (f1["bn.running_var"] >= 0.5).all()


def test_compat_surface_for_data_loaders_and_metrics():
    """`from libdl.data_loaders import ...` / `from libdl.metrics import ...` of the experiment scripts
    (exp180d...py:17-18) resolve through the compat shim."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(1, %r);"
            "from libdl.data_loaders import dataset_context, dataset_context_segm, dataset_context_measuresegm;"
            "from libdl.metrics import early_stopping, calculate_eval_measures, calculate_single_measure, "
            "calculate_mpe_measures_mireval;"
            "import multipitch_architectures_amd.data_loaders as d;"
            "assert dataset_context is d.dataset_context; print('ok')"
            % (os.path.join(ROOT, "multipitch_architectures_amd", "compat"), ROOT))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_early_stopping_monitor():
    from multipitch_architectures_amd.metrics.monitoring import early_stopping
    es = early_stopping(mode="max", min_delta=1e-5, patience=2)
    assert [es.step(v) for v in (0.5, 0.6, 0.6, 0.59)] == [False, False, False, True]
    es = early_stopping(mode="min", min_delta=10, patience=1, percentage=True)
    assert [es.step(v) for v in (1.0, 0.95)] == [False, True]         # 5 % is not a 10 % improvement
    assert es.curr_is_better(0.89) and not es.curr_is_better(0.91)
    es = early_stopping(patience=0)
    assert [es.step(v) for v in (1.0, 2.0, float("nan"))] == [False, False, False]
    es = early_stopping(patience=5)
    assert [es.step(v) for v in (1.0, float("nan"))] == [False, True]
    with pytest.raises(ValueError):
        early_stopping(mode="median")


def test_data_and_metrics_have_no_cpu_path():
    import numpy as np
    from multipitch_architectures_amd.data_loaders import dataset_context
    from multipitch_architectures_amd.metrics import calculate_eval_measures
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        dataset_context(np.zeros((6, 100, 216), np.float32), np.zeros((100, 72), np.float32),
                        {"context": 75, "stride": 1, "compression": 10})
    with pytest.raises(RuntimeError):
        calculate_eval_measures(np.zeros((4, 72), np.float32), np.zeros((4, 72), np.float32), ["precision"])


def test_planner_invariants_on_random_geometries():
    """The planners are host code: for every stride-1 (and the head's stride == kernel width) geometry they must produce
    a plan whose LDS footprint fits a CU, a positive packed-filter size for both pack modes and a positive workspace."""
    import ctypes
    import random
    from multipitch_architectures_amd import _lib as L
    lib = L.load()
    rnd = random.Random(7)
    seen = 0
    for _ in range(300):
        kh, kw = rnd.choice([(1, 1), (3, 3), (5, 5), (9, 9), (15, 15), (75, 1), (1, 61), (2, 5), (3, 1)])
        sw = 3 if (kw == 3 and rnd.random() < 0.3) else 1
        ph, pw = (kh // 2, 0 if sw == 3 else kw // 2) if rnd.random() < 0.7 else (0, 0)
        H, W = rnd.randint(kh, 180), rnd.randint(max(kw, 3), 220)
        if sw == 3:
            W -= W % 3
            if W < 3:
                continue
        B, Cin, Cout = rnd.choice([1, 2, 25, 32, 50, 256]), rnd.choice([1, 4, 6, 16, 30, 70, 128, 256]), rnd.choice([1, 8, 16, 50, 70, 128])
        d = L.ConvDesc(B, Cin, H, W, Cout, kh, kw, 1, sw, ph, pw)
        for mode in (0, 1, 2):
            buf = ctypes.create_string_buffer(512)
            rc = lib.mpa_conv2d_describe_plan(ctypes.byref(d), mode, buf, 512)
            assert rc == 0, (d.key(), mode, rc)
            text = buf.value.decode()
            lds = int(re.search(r"lds=(\d+)B", text).group(1))
            # (the head conv2 kernels of conv_head.hip run one workgroup per CU on up to 150 KB; everything else two or more)
            assert 0 < lds <= (150 if text.startswith(("head_", "tall_")) else 80) * 1024, text
        assert lib.mpa_conv2d_packed_floats(ctypes.byref(d), 0) > 0
        assert lib.mpa_conv2d_packed_floats(ctypes.byref(d), 1) > 0
        assert lib.mpa_conv2d_bwd_weight_workspace(ctypes.byref(d)) > 0
        seen += 1
    assert seen > 250


def test_bench_spawns_its_own_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` with no launcher: the parent starts two rank processes (never touching a GPU itself)
    and returns non-zero when they fail -- here because this box has no MI355X and there is no CPU fallback."""
    import subprocess
    import sys
    if torch.cuda.device_count() > 0:
        pytest.skip("would start a real 2-rank run on this box; covered by tests/test_gpu_parallel.py there")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert r.stderr.count("needs an MI355X") >= 1 and "stopping the other ranks" in r.stderr
    assert r.stdout.strip() == ""


def test_adamw_load_state_dict_and_add_param_group_drop_the_device_tables():
    """the update kernel reads step count / exp_avg pointers from device tables built lazily from ``opt.state``: loading a
    state dict (resume) or adding a group must invalidate them and tell ``TrainStep`` that its graph is stale"""
    import torch
    from multipitch_architectures_amd.optim import AdamW
    w = torch.nn.Parameter(torch.zeros(3))
    opt = AdamW([w], lr=1e-3)
    e0 = opt.table_epoch
    opt._tables[0] = {"key": None, "gkey": (1,)}
    opt.invalidate_grad_table()
    assert opt._tables[0]["gkey"] is None
    opt.load_state_dict(opt.state_dict())
    assert opt._tables == {} and opt.table_epoch == e0 + 1
    opt.add_param_group({"params": [torch.nn.Parameter(torch.zeros(2))]})
    assert opt.table_epoch == e0 + 2


def test_variant_classes_have_the_reference_state_dict_schema():
    """the 8 re-composed U-Net variants: key names, order and shapes as recorded from the reference classes"""
    import json
    import numpy as np
    from multipitch_architectures_amd import nn_models
    from multipitch_architectures_amd.configs import VARIANT_CONFIGS
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, kw in VARIANT_CONFIGS.items():
        g = np.load(os.path.join(gdir, f"xcls-{name}.npz"))
        want = json.loads(str(g["schema"]))
        got = {k: list(v.shape) for k, v in getattr(nn_models, name)(**kw).state_dict().items()}
        assert list(got) == list(want) and got == want, name
        sig = [[q.name, q.default] for q in list(inspect.signature(getattr(nn_models, name).__init__).parameters.values())[1:]]
        assert sig == json.loads(str(g["signature"])), name           # constructor keywords / defaults verbatim


def test_hcqt_harmonics_are_grouped_by_octave_classes():
    """compute_efficient_hcqt's sharing rule (hcqt.py:130-152): planes a whole number of octaves apart are slices of one CQT"""
    from multipitch_architectures_amd.data_preprocessing.hcqt import _octave_classes
    # the paper's setting, planes = [1/2, 1, 2, 3, 4, 5]: one transform from the subharmonic up carries 1/2, 1, 2 and 4
    assert _octave_classes(5, 1) == [(0.5, [(0, 0), (1, 1), (2, 2), (3, 4)]), (3.0, [(0, 3)]), (5.0, [(0, 5)])]
    assert _octave_classes(3, 0) == [(1.0, [(0, 0), (1, 1)]), (3.0, [(0, 2)])]
    # planes [1/3, 1/2, 1, 2, 3, 4, 5, 6]: 1/3 stands alone (3 is two octaves above 3/4, not above 1/3); 6 = 2 * 3
    assert _octave_classes(6, 2) == [(1 / 3, [(0, 0)]), (0.5, [(0, 1), (1, 2), (2, 3), (3, 5)]), (3.0, [(0, 4), (1, 7)]),
                                     (5.0, [(0, 6)])]
    for h, s in ((5, 1), (8, 3), (1, 0)):
        planes = sorted(p for _, members in _octave_classes(h, s) for _, p in members)
        assert planes == list(range(h + s))
