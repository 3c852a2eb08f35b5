// microbenchmark: groups of 96 MFMAs (4x8 accumulators, 3 taps) preceded by NV VALU instructions that produce the operands
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[3][4], b[3][8];
  for (int u = 0; u < 3; ++u) { for (int i = 0; i < 4; ++i) a[u][i] = threadIdx.x * 1e-3f + i; for (int j = 0; j < 8; ++j) b[u][j] = 1.f - j; }
  for (int it = 0; it < iters; ++it) {
    if (NV > 0) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) a[u][i] = (float)(it + u + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) b[u][j] = (float)(it - u - j);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NV>
void run(int blocks, int iters) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NV>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 96 * 2048.0;
    if (rep == 2) printf("NV %d blocks %5d: %.3f ms  %.1f TFLOP/s\n", NV, blocks, ms, flop / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<0>(512, 8000); run<1>(512, 8000); run<1>(256, 8000); run<1>(768, 8000);
  return 0;
}
