// microbenchmark: <1,12>-shaped groups: G taps x (1 A + 12 B) operands produced by VALU, then G*12 MFMAs
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int G, int LDSB>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  extern __shared__ float lds[];
  f32x4 acc[12];
  for (int j = 0; j < 12; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[G], b[G][12];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < G; ++u) {
      a[u] = (float)(it + u);
#pragma unroll
      for (int j = 0; j < 12; ++j) b[u][j] = (float)(it - u - j);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; ++u)
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u][j], acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int j = 0; j < 12; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  if (LDSB > 0 && threadIdx.x == 0) lds[0] = s;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int G>
void run(int blocks, int iters, int ldsb) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipFuncSetAttribute((const void*)k<G, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<G, 1>), dim3(blocks), dim3(256), ldsb, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * G * 12 * 2048.0;
    if (rep == 2) printf("G %d blocks %5d lds %6d: %.3f ms  %.1f TFLOP/s\n", G, blocks, ldsb, ms, flop / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<3>(768, 20000, 50 * 1024);   // 3 workgroups per CU
  run<3>(512, 20000, 70 * 1024);   // 2 per CU
  run<3>(256, 20000, 150 * 1024);  // 1 per CU
  run<1>(768, 60000, 50 * 1024);
  run<6>(768, 10000, 50 * 1024);
  return 0;
}
