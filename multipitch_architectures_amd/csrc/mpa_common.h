// Shared helpers for the gfx950 kernels of libmpa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mpa.h"
#include "mpa_diag.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MPA_WAVE 64

// PyTorch can leave a stale (non-fatal) error in the thread's HIP error slot; clear it before every launch so that
// mpa_launch_status() reports this launch only.
#define MPA_LAUNCH(...)            \
  do {                             \
    (void)hipGetLastError();       \
    hipLaunchKernelGGL(__VA_ARGS__); \
  } while (0)

static inline int mpa_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MPA_OK : MPA_ERR_LAUNCH;
}

static inline int64_t mpa_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Zero-fill as an ordinary kernel (rows x row_words 4-byte words, row pitch in words).  Used instead of
// hipMemsetAsync / hipMemset2DAsync everywhere: memset *nodes* inside a captured HIP graph were replayed out of order
// on ROCm 7.2 whenever the stream had seen an un-synchronised kernel before the graph launch (deterministic garbage in
// the split-K accumulators); a kernel node is ordered like every other kernel of the graph.
__global__ void mpa_zero_kernel(uint32_t* __restrict__ p, long rows, long row_words, long pitch_words);
int mpa_zero_async(void* ptr, size_t bytes, hipStream_t s);
int mpa_zero2d_async(void* ptr, size_t pitch_bytes, size_t width_bytes, size_t rows, hipStream_t s);

__device__ __forceinline__ float mpa_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double mpa_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float mpa_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float mpa_apply_act(float v, int act, float slope) {
  switch (act) {
    case MPA_ACT_RELU: return v > 0.f ? v : 0.f;
    case MPA_ACT_LRELU: return v >= 0.f ? v : v * slope;
    case MPA_ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
    case MPA_ACT_ELU: return v > 0.f ? v : expm1f(v);
    default: return v;
  }
}
