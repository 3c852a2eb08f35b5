// microbenchmark: does the VGPR bank of srcA / srcB matter for v_mfma_f32_16x16x4_f32 throughput?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MODE 0: 4 A regs x 8 B regs, all combinations (like conv_fwd_kernel<4,8>)
// MODE 1: 1 A reg x 8 B regs
// MODE 2: 4 A regs x 8 B regs but B values broadcast from ONE register (same reg for all)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[4][8];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[4], b[8];
  for (int i = 0; i < 4; ++i) a[i] = threadIdx.x * 1e-3f + i;
  for (int j = 0; j < 8; ++j) b[j] = threadIdx.x * 2e-3f - j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float av = MODE == 1 ? a[0] : a[i];
        const float bv = MODE == 2 ? b[0] : b[j];
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i][j], 0, 0, 0);
      }
    // keep the operands "live and changing" without real cost
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
    asm volatile("" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
void run(int blocks, int iters) {
  float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE>), dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)blocks * 4 * iters * 32 * 2048.0;
    if (rep == 2) printf("MODE %d blocks %5d: %.3f ms  %.1f TFLOP/s\n", MODE, blocks, ms, flop / ms / 1e9);
  }
  hipFree(out);
}
int main() {
  run<0>(512, 25000); run<1>(512, 25000); run<2>(512, 25000);
  return 0;
}
