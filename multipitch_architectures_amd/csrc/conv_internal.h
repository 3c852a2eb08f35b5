// Entry points the convolution translation units call across each other (C++ linkage, not part of the C ABI).
#pragma once
#include "mpa_common.h"

// generic (non-15x15) backward-weight: conv_wgrad.hip
int mpa_conv_wgrad_generic(const mpa_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* workspace,
                           int64_t workspace_bytes, hipStream_t s);
// deterministic sum of the per-slice partial results ws [S][Cout][NtotP] -> dw [Cout][Ntot] (+ db from the last column)
int mpa_conv_reduce_partials(const float* ws, float* dw, float* db, int Cout, int Ntot, int NtotP, int S, hipStream_t s);
