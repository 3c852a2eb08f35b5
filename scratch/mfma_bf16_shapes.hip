// round 3 microbenchmark: sustained rate of the two gfx950 bf16 MFMA shapes in the issue patterns the split-bf16 (3-product)
// convolution uses -- three dependent MFMAs per accumulator (hi*hi, hi*lo, lo*hi), NACC accumulators round-robin.
// build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form scratch/mfma_bf16_shapes.hip -o scratch/mfma_bf16_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int KIND, int CHAIN>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters) {
  float s = 0.f;
  bf16x8 a[3], b[3];
  for (int u = 0; u < 3; ++u)
    for (int j = 0; j < 8; ++j) {
      a[u][j] = (__bf16)(in[(threadIdx.x * 8 + j + u * 17) & 1023]);
      b[u][j] = (__bf16)(in[(threadIdx.x * 8 + j + u * 29 + 5) & 1023]);
    }
  if constexpr (KIND == 1) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[c % 3], b[(c + 1) % 3], acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  } else {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int c = 0; c < CHAIN; ++c) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[c % 3], b[(c + 1) % 3], acc[i], 0, 0, 0);
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][15];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int KIND, int CHAIN>
void run(int blocks, int threads, int iters, const float* in, const char* name) {
  float* out; hipMalloc(&out, (size_t)blocks * threads * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double flop_per = KIND == 1 ? 2.0 * 16 * 16 * 32 : 2.0 * 32 * 32 * 16;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, KIND, CHAIN>), dim3(blocks), dim3(threads), 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * (threads / 64) * iters * NACC * CHAIN * flop_per;
    if (rep == 2) printf("%-14s NACC %2d chain %d blocks %4d x %3d thr: %8.3f ms  %8.1f TFLOP/s (3-product equivalent %6.1f)\n", name, NACC,
                         CHAIN, blocks, threads, ms, flop / ms / 1e9, flop / ms / 1e9 / 3);
  }
  hipFree(out);
}
int main() {
  float h[1024];
  unsigned r = 12345;
  for (int i = 0; i < 1024; ++i) { r = r * 1664525u + 1013904223u; h[i] = ((r >> 8) & 0xffff) / 32768.f - 1.f; }
  float* in; hipMalloc(&in, sizeof(h)); hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  run<8, 1, 1>(256, 256, 40000, in, "16x16x32");
  run<8, 1, 3>(256, 256, 15000, in, "16x16x32");
  run<15, 1, 3>(256, 256, 8000, in, "16x16x32");
  run<8, 1, 3>(256, 512, 15000, in, "16x16x32");
  run<8, 1, 3>(512, 256, 15000, in, "16x16x32");
  run<4, 2, 1>(256, 256, 40000, in, "32x32x16");
  run<4, 2, 3>(256, 256, 15000, in, "32x32x16");
  run<4, 2, 3>(256, 512, 15000, in, "32x32x16");
  return 0;
}
