"""The convolution planners are host-only C++ (csrc/conv_plan.h, split out of the kernel files in round 3): this test
compiles them with g++ alone -- no hipcc, no GPU -- and checks the plans of the models' layers: every plan exists, fits the
LDS budget, its tiles cover the output, and the choices DESIGN.md quotes are the ones made."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def plans(tmp_path_factory):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path_factory.mktemp("plan") / "plan_driver")
    subprocess.run([gxx, "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "cpp", "plan_driver.cpp")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    res = {}
    for ln in out.splitlines():
        name, kind, rest = ln.split(" ", 2)
        res[(name, kind)] = {k: int(v) for k, v in re.findall(r"(\w+)=(-?\d+)", rest)}
    return res


def test_every_layer_has_a_plan_that_fits(plans):
    for (name, kind), p in plans.items():
        if name == "strided_unsupported":
            continue
        assert p["ok"] == 1, (name, kind)
        assert p["lds"] <= 160 * 1024, (name, kind, p)
        if kind in ("fwd", "dgrad"):
            assert p["tilesY"] * p["TH"] >= p["OH"] and p["tilesX"] * p["TW"] >= p["OW"], (name, kind, p)
            assert p["TH"] * p["TW"] <= p["PB"] * 64, (name, kind, p)             # a tile fits the block's pixel lanes
            assert p["coTiles"] * p["COT"] >= 1


def test_unsupported_geometry_is_reported_not_planned(plans):
    assert plans[("strided_unsupported", "dgrad")]["ok"] == 0      # stride (2,2) has no backward-data decomposition


def test_the_choices_design_md_quotes(plans):
    f = plans[("upconv4b", "fwd")]
    assert (f["NB"], f["PB"], f["KWS"], f["KS"]) == (2, 12, 15, 1)      # conv_fwd_kernel<2,12,15>: the roofline kernel
    d = plans[("upconv4b", "dgrad")]
    assert (d["NB"], d["PB"], d["KWS"]) == (1, 12, 15)
    assert plans[("upconv4b_b32", "dgrad")]["KS"] > 1                   # small grid: channel slices level the tile count
    w = plans[("upconv4b", "wgrad15")]
    assert w["ga"] == 1 and w["n32"] == 4 and w["fold"] == 0
    p = plans[("prefilt", "wgrad15")]
    assert (p["n32"], p["has16"], p["fold"]) == (2, 0, 6)               # 70 couts = 2 x 32 + tap-folded 6
    assert plans[("conv2_80", "wgrad")]["NBC"] == 5                     # head conv2 on the generic kernels: 80-cout block shape


def test_head_conv2_and_fold_plans(plans):
    """csrc/conv_head.hip (the head's 3x3 stride-(1,3) conv2) and the cout remainder fold of the 15x15 layers"""
    f, b, w = plans[("conv2_80", "head0")], plans[("conv2_80", "head1")], plans[("conv2_80", "headwg")]
    assert (f["MT"], f["WM"], f["WN"], f["NT"], f["CK"]) == (5, 1, 4, 9, 4)      # 5 cout tiles x 5 pixel blocks per wave, 2 WGs / CU
    assert f["PXT"] * f["tilesP"] >= f["P"] and f["XS"] % 32 == 16 and f["lds"] <= 78 * 1024
    assert (b["MT"], b["WM"], b["WN"], b["NT"], b["rows"]) == (6, 4, 2, 3, 384)  # rows = (channel, column phase) pairs
    assert b["XS"] % 32 == 16
    assert (w["MT"], w["coGroups"], w["chGroups"], w["NCS"], w["SEG"], w["NRB"]) == (5, 1, 2, 2, 36, 1)
    assert w["S"] * w["chGroups"] * w["coGroups"] == 256                         # one workgroup per CU
    f2, w2 = plans[("conv2_200", "head0")], plans[("conv2_200", "headwg")]
    assert (f2["MT"], f2["WM"]) == (7, 2) and (w2["MT"], w2["coGroups"]) == (5, 3)
    assert ("conv2_80_b16", "head0") not in plans and ("conv2_80_b32", "head0") in plans   # small launch: the generic forward kernel
    assert plans[("conv2_80_b32", "headwg")]["NRB"] == 2                         # row blocks give 128 slices their items
    for k in ("foldf", "foldb"):
        p = plans[("prefilt", k)]
        assert (p["C0"], p["R"], p["V"]) == (64, 6, 2)                           # 70 = 64 + 6 rows x 2
    assert ("upconv4b", "foldf") not in plans and ("conv2_80", "foldf") not in plans


def test_tall_filter_plans(plans):
    """conv3's (75,1) filters at T > 75 (conv_plan.h: plan_tall): tap groups of 25 on the head GEMM kernel; at T = 75 the layer is a
    GEMM (ops.conv2d) and no tall plan exists"""
    f, b = plans[("conv3_T174", "tall0")], plans[("conv3_T174", "tall1")]
    assert (f["MT"], f["WM"], f["NT"], f["NG"], f["rows"], f["P"]) == (4, 1, 25, 3, 50, 100 * 72)
    assert (b["MT"], b["WM"], b["NT"], b["NG"], b["rows"], b["P"]) == (5, 1, 25, 3, 80, 174 * 72)
    assert f["XS"] % 32 == 16 and f["lds"] <= 150 * 1024 and b["lds"] <= 150 * 1024
    assert f["chunks"] == 20 * 3 and b["chunks"] == 13 * 3
    assert ("conv3_T75", "tall0") not in plans and ("conv3_T75", "tall1") not in plans
