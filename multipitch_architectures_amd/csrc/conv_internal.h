// Entry points the convolution translation units call across each other (C++ linkage, not part of the C ABI).
#pragma once
#include "mpa_common.h"

// generic (non-15x15) backward-weight: conv_wgrad.hip
int mpa_conv_wgrad_generic(const mpa_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* workspace,
                           int64_t workspace_bytes, hipStream_t s);
// deterministic sum of the per-slice partial results ws [S][Cout][NtotP] -> dw [Cout][Ntot] (+ db from the last column)
int mpa_conv_reduce_partials(const float* ws, float* dw, float* db, int Cout, int Ntot, int NtotP, int S, hipStream_t s);

// head conv2 (3x3, stride (1,3), padding (1,0)): conv_head.hip.  MPA_ERR_UNSUPPORTED when conv_plan.h's plan_head /
// plan_head_wgrad does not take the problem (the generic kernels do then).
int mpa_conv_head_fwd(const mpa_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, float slope,
                      hipStream_t s);
int mpa_conv_head_bwd_data(const mpa_conv_desc* d, const float* dy, const float* wp, float* dx, hipStream_t s);
int64_t mpa_conv_head_wgrad_workspace(const mpa_conv_desc* d);
int mpa_conv_head_bwd_weight(const mpa_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* workspace,
                             int64_t workspace_bytes, hipStream_t s);
// tall (kh,1) filters (conv3 at T > 75) on the same GEMM kernel: conv_head.hip, conv_plan.h: plan_tall
int mpa_conv_tall_fwd(const mpa_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, float slope,
                      hipStream_t s);
int mpa_conv_tall_bwd_data(const mpa_conv_desc* d, const float* dy, const float* wp, float* dx, hipStream_t s);
