"""ctypes binding of libmpa_hip.so (C ABI declared in include/mpa.h).

The product path has no CPU fallback: if the library is missing or a tensor is
not a contiguous fp32 HIP tensor, the call raises.
"""
import ctypes
import os

import torch  # noqa: F401  -- must be imported BEFORE libmpa_hip.so is loaded: the library has to bind to the same
#                        libamdhip64 that PyTorch brings, otherwise streams/pointers belong to another HIP runtime

from .build import DIAG_LIB, LIB, build_library

c_void_p, c_int, c_int64, c_float, c_double, c_uint64 = (
    ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double, ctypes.c_uint64)


class ConvDesc(ctypes.Structure):
    """mirror of struct mpa_conv_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "Cin", "H", "W", "Cout", "kh", "kw", "sh", "sw", "ph", "pw")]

    def key(self):
        return tuple(getattr(self, n) for n, _ in self._fields_)

    @property
    def OH(self):
        return (self.H + 2 * self.ph - self.kh) // self.sh + 1

    @property
    def OW(self):
        return (self.W + 2 * self.pw - self.kw) // self.sw + 1


_P = c_void_p
class ContextDesc(ctypes.Structure):
    """mirror of struct mpa_context_desc"""
    _fields_ = [(n, ctypes.c_int32) for n in ("n_harm", "n_bins", "frames", "n_out", "seglength", "flags")] + \
               [("compression", ctypes.c_float), ("noisestd", ctypes.c_float)]


CTX_EQ, CTX_NOISE, CTX_LOG, CTX_TUNE, CTX_TRANSP, CTX_SEGM_TARGETS = 1, 2, 4, 8, 16, 32

_D = ctypes.POINTER(ConvDesc)

# name -> (restype, argtypes); must list every symbol include/mpa.h declares
SIGNATURES = {
    "mpa_strerror": (ctypes.c_char_p, [c_int]),
    "mpa_version": (c_int, []),
    "mpa_diag_reload": (c_int, []),
    "mpa_conv2d_packed_floats": (c_int64, [_D, c_int]),
    "mpa_conv2d_pack": (c_int, [_D, c_int, _P, _P, _P]),
    "mpa_conv2d_pack_entry_bytes": (c_int, []),
    "mpa_conv2d_pack_entry": (c_int, [_D, c_int, _P, _P, _P]),
    "mpa_conv2d_pack_many": (c_int, [_P, c_int, _P]),
    "mpa_conv2d_fwd": (c_int, [_D, _P, _P, _P, _P, c_int, c_float, _P]),
    "mpa_conv2d_fold_supported": (c_int, [_D]),
    "mpa_conv2d_fwd_folded": (c_int, [_D, _P, _P, _P, _P, c_int, c_float, _P]),
    "mpa_conv2d_fwd_stats_rows": (c_int64, [_D]),
    "mpa_conv2d_fwd_stats": (c_int, [_D, _P, _P, _P, _P, _P, _P]),
    "mpa_conv2d_bwd_data": (c_int, [_D, _P, _P, _P, _P]),
    "mpa_conv2d_describe_plan": (c_int, [_D, c_int, ctypes.c_char_p, c_int]),
    "mpa_conv2d_bwd_weight_workspace": (c_int64, [_D]),
    "mpa_conv2d_bwd_weight": (c_int, [_D, _P, _P, _P, _P, _P, c_int64, _P]),
    "mpa_bf16x3_split_bytes": (c_int64, [c_int, c_int, c_int, c_int]),
    "mpa_bf16x3_split": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "mpa_conv2d_bf16x3_supported": (c_int, [_D, c_int]),
    "mpa_conv2d_bf16x3_packed_bytes": (c_int64, [_D, c_int]),
    "mpa_conv2d_bf16x3_pack": (c_int, [_D, c_int, _P, _P, _P]),
    "mpa_conv2d_bf16x3_stats_rows": (c_int64, [_D]),
    "mpa_conv2d_bf16x3_fwd": (c_int, [_D, _P, _P, _P, _P, c_int, c_float, _P, _P]),
    "mpa_conv2d_bf16x3_bwd_data": (c_int, [_D, _P, _P, _P, _P]),
    "mpa_conv2d_bf16x3_bwd_weight_workspace": (c_int64, [_D]),
    "mpa_conv2d_bf16x3_bwd_weight": (c_int, [_D, _P, _P, _P, _P, _P, c_int64, _P]),
    "mpa_layernorm_cf_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, _P]),
    "mpa_layernorm_bwd_workspace": (c_int64, [c_int]),
    "mpa_layernorm_cf_bwd_ws": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mpa_layernorm_rows_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_float, _P]),
    "mpa_layernorm_rows_bwd_ws": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int, _P]),
    "mpa_bn_relu_train_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, _P]),
    "mpa_bn_relu_train_fwd_partials": (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, _P]),
    "mpa_bn_relu_eval_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_int, _P]),
    "mpa_bn_relu_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "mpa_bn_batch_sums": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mpa_bn_relu_train_fwd_sums": (c_int, [_P, _P, c_double, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, _P]),
    "mpa_bn_relu_bwd_sums": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mpa_bn_relu_bwd_apply": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_double, _P, c_int, c_int, c_int, c_int, _P]),
    "mpa_attn_batchaxis_fwd_kv": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "mpa_attn_batchaxis_bwd_kv": (c_int, [_P] * 9 + [c_int, c_int, c_int, c_int, c_int, _P]),
    "mpa_maxpool2d_fwd": (c_int, [_P, _P, _P] + [c_int] * 10 + [_P]),
    "mpa_maxpool2d_bwd": (c_int, [_P, _P, _P] + [c_int] * 10 + [_P]),
    "mpa_maxunpool2d_fwd": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P]),
    "mpa_maxunpool2d_bwd": (c_int, [_P, _P, _P] + [c_int] * 6 + [_P]),
    "mpa_maxpool2d_bwd_add": (c_int, [_P, _P, _P, c_int64, _P] + [c_int] * 10 + [_P]),
    "mpa_upcat_fwd": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P]),
    "mpa_upcat_bwd": (c_int, [_P, _P, _P] + [c_int] * 7 + [_P]),
    "mpa_upcat_scaled_fwd": (c_int, [_P, _P, _P] + [c_int] * 9 + [_P]),
    "mpa_upcat_scaled_bwd": (c_int, [_P, _P, _P] + [c_int] * 9 + [_P]),
    "mpa_logsoftmax_cat_fwd": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "mpa_logsoftmax_cat_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "mpa_act_fwd": (c_int, [_P, _P, c_int64, c_int, c_float, _P]),
    "mpa_act_bwd": (c_int, [_P, _P, _P, c_int64, c_int, c_float, _P]),
    "mpa_dropout": (c_int, [_P, _P, c_int64, c_float, _P, c_uint64, _P]),
    "mpa_poolrows_dropout_add_fwd": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_float, _P, c_uint64, _P]),
    "mpa_poolrows_dropout_bwd": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, c_float, _P, c_uint64, _P]),
    "mpa_poolrows_dropout_act_bwd": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, c_float, _P, c_uint64, c_float, _P]),
    "mpa_u64_add": (c_int, [_P, c_uint64, _P]),
    "mpa_gather_copy": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "mpa_store_ptrs": (c_int, [_P, _P, c_int, _P]),
    "mpa_add": (c_int, [_P, _P, _P, c_int64, _P]),
    "mpa_axpy": (c_int, [c_float, _P, _P, c_int64, _P]),
    "mpa_scale": (c_int, [c_float, _P, c_int64, _P]),
    "mpa_scale_by": (c_int, [_P, _P, _P, c_int64, _P]),
    "mpa_transpose_add": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mpa_channel_sum": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "mpa_add_rows_bcast": (c_int, [_P, _P, _P, c_int, c_int64, _P]),
    "mpa_gemm": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P]),
    "mpa_gemm_masked": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, c_int, c_int, c_int, _P]),
    "mpa_gemm_batched": (c_int, [c_int, _P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P]),
    "mpa_gemm_bf16x3_supported": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, c_int, c_int, c_int]),
    "mpa_gemm_bf16x3": (c_int, [_P, c_int64, c_int64, _P, c_int64, c_int64, _P, _P, c_int64, c_int, c_int, c_int, c_int, c_int, _P]),
    "mpa_colsum": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "mpa_attn_batchaxis_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "mpa_attn_batchaxis_bwd": (c_int, [_P] * 9 + [c_int, c_int, c_int, c_int, _P]),
    "mpa_lstm_cell_fwd": (c_int, [_P, c_int64, _P, _P, _P, c_int64, _P, c_int, c_int, _P]),
    "mpa_lstm_cell_bwd": (c_int, [_P, c_int64, _P, _P, _P, _P, _P, _P, c_int64, _P, c_int, c_int, _P]),
    "mpa_bce_fwd": (c_int, [_P, _P, _P, c_int64, _P]),
    "mpa_bce_bwd": (c_int, [_P, _P, _P, c_int64, _P, _P]),
    "mpa_ce_fwd_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "mpa_time_scale": (c_int, [_P, ctypes.c_int64, c_int, c_int, c_int, c_int, c_int, _P, _P]),
    "mpa_context_batch": (c_int, [ctypes.POINTER(ContextDesc), c_int, _P, _P, _P, _P, _P, _P, _P, ctypes.c_uint64, _P, _P, _P]),
    "mpa_eval_measures_workspace": (c_int64, [c_int64, c_int]),
    "mpa_eval_measures": (c_int, [_P, _P, c_int64, c_int, c_double, _P, _P, c_int64, _P]),
    "mpa_annotation_workspace": (c_int64, [c_int]),
    "mpa_annotation_array_nooverlap": (c_int, [_P, c_int, c_int, c_double, c_double, c_int, c_int, _P, _P, c_int64, _P]),
    "mpa_reflect_pad": (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    "mpa_stft_basis": (c_int, [_P, c_int, _P]),
    "mpa_complex_mag": (c_int, [_P, _P, c_int64, _P]),
    "mpa_piptrack": (c_int, [_P, c_int64, c_int, c_double, c_int, c_double, c_double, c_double, _P, _P, _P, _P]),
    "mpa_pitch_tuning_workspace": (c_int64, [c_int64]),
    "mpa_pitch_tuning": (c_int, [_P, _P, c_int64, c_int, c_double, _P, _P, c_int64, _P]),
    "mpa_cqt_basis": (c_int, [_P, c_int64, c_int64, c_int, c_double, c_int, c_int, c_double, _P]),
    "mpa_cqt_mag_scatter": (c_int, [_P, c_int64, c_int, c_int, c_int, _P, c_int, c_int, _P, _P, c_int, _P]),
    "mpa_adamw_step": (c_int, [_P, _P, _P, _P, _P, c_int, c_int64, _P, c_double, c_double, c_double, c_double, _P]),
}

_lib = None


def load(build_if_missing: bool = False):
    """Return the loaded library; raise (never fall back) when it cannot be loaded."""
    global _lib
    if _lib is not None:
        return _lib
    diag = os.environ.get("MPA_DIAG_LIB", "0") == "1"      # scratch/ timing experiments: the -DMPA_DIAG build
    path = DIAG_LIB if diag else LIB
    if build_if_missing:
        path = build_library(diag=diag)          # no-op when the in-tree .so matches the sources' digest
    if not os.path.exists(path):
        if True:
            raise RuntimeError(
                f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(multipitch_architectures_amd has no CPU/PyTorch fallback for its kernels)")
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class MpaError(RuntimeError):
    pass


def check(rc, what):
    if rc is None:
        return
    if rc < 0:
        msg = load().mpa_strerror(int(rc)).decode()
        raise MpaError(f"{what} failed: {msg} (code {rc})")
    return rc
