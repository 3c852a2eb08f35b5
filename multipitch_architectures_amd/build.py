"""Build libmpa_hip.so (the gfx950 kernel library) in-tree with hipcc.

Objects are compiled in parallel and linked into
``multipitch_architectures_amd/csrc/libmpa_hip.so`` so that the built library
travels with the repo snapshot to the GPU box.  hipcc cross-compiles for gfx950
without a GPU present.
"""
import concurrent.futures
import hashlib
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libmpa_hip.so")
SOURCES = ["conv_fwd_nb12.hip", "conv_fwd_nb45.hip", "conv_fwd_nb36.hip", "conv_fwd.hip", "conv_wgrad.hip", "conv_wgrad15.hip", "conv_head.hip", "conv_bf16x3.hip", "norm.hip", "pool_up.hip", "pointwise.hip", "gemm.hip", "attn.hip", "data.hip", "metrics.hip", "annot.hip", "hcqt.hip"]
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs.  Without it hipcc (ROCm 7.2) parks small accumulator
# sets in VGPRs and round-trips them through a single AGPR tuple around every v_mfma (8 v_accvgpr moves + s_nop 9
# per MFMA in conv_fwd_kernel<1,6>).
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-pass-failed", "-ffp-contract=off",
         "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libmpa_hip.so")
    return exe


def _digest():
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for name in sorted(os.listdir(CSRC)) + ["../../include/mpa.h"]:
        if name.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()


DIAG_LIB = os.path.join(CSRC, "libmpa_hip_diag.so")


def build_library(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """diag=True: the -DMPA_DIAG build (kernel-side debug switches of csrc/mpa_diag.h compiled in) as libmpa_hip_diag.so,
    for the timing experiments under scratch/ -- never what the package loads by default."""
    lib = DIAG_LIB if diag else LIB
    flags = FLAGS + (["-DMPA_DIAG"] if diag else [])
    stamp = lib + ".stamp"
    digest = _digest() + ("-diag" if diag else "")
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == digest:
        return lib
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build_diag" if diag else "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr:
            print(r.stderr)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(digest)
    return lib


if __name__ == "__main__":
    import sys
    print(build_library(force=True, verbose=True, diag="--diag" in sys.argv))
