// microbenchmark for the split-bf16 option: sustained v_mfma_f32_16x16x16_bf16 and
// v_mfma_f32_32x32x8_bf16 rates against the fp32-input v_mfma_f32_16x16x4_f32 the convolutions use today.
// A 3-pass split (a_hi b_hi + a_hi b_lo + a_lo b_hi) does three bf16 MFMAs per fp32-equivalent product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int NACC, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float s = 0.f;
  if constexpr (KIND == 0) {          // fp32 16x16x4
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = threadIdx.x * 2e-3f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  } else if constexpr (KIND == 1) {   // bf16 16x16x32 (gfx950: 8 bf16 per lane)
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(threadIdx.x * 2e-3f - j); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  } else {                            // bf16 32x32x16
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 1e-3f + j); b[j] = (__bf16)(threadIdx.x * 2e-3f - j); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC, int KIND>
void run(int blocks, int iters, double flop_per_mfma, const char* name) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, KIND>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * NACC * flop_per_mfma;
    if (rep == 2) printf("%-22s NACC %2d blocks %4d: %8.3f ms  %8.1f TFLOP/s  (3-pass fp32-equivalent %7.1f)\n", name, NACC, blocks, ms,
                         flop / ms / 1e9, KIND ? flop / ms / 1e9 / 3 : flop / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<8, 0>(512, 50000, 2.0 * 16 * 16 * 4, "f32 16x16x4");
  run<8, 1>(512, 50000, 2.0 * 16 * 16 * 32, "bf16 16x16x32");
  run<4, 2>(512, 50000, 2.0 * 32 * 32 * 16, "bf16 32x32x16");
  run<8, 1>(256, 50000, 2.0 * 16 * 16 * 32, "bf16 16x16x32");
  run<4, 2>(256, 50000, 2.0 * 32 * 32 * 16, "bf16 32x32x16");
  return 0;
}
