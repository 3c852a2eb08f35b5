// does v_mfma_f32_16x16x32_bf16 round its accumulation to nearest, or truncate?  acc0 = 2^22 (ulp 0.5), every MFMA adds
// 32 * 1 * (1 + 3/128) = 32.75: round-to-nearest-even gives +32.75 per step on average, truncation +32.5.
// Second experiment: products that need more than the accumulator's ulp *inside* one MFMA (32 products of 2^-10 each
// against acc 2^14: their sum 2^-5 * ... ) -- are the 32 products summed exactly before the one rounding?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k32(float* out, float acc0, float bval, int n) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)bval; }
  f32x16 acc;
  for (int j = 0; j < 16; ++j) acc[j] = acc0;
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void k(float* out, float acc0, float bval, int n) {
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)1.0f; b[j] = (__bf16)bval; }
  f32x4 acc = {acc0, acc0, acc0, acc0};
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = acc[0];
}
int main() {
  float* out; hipMalloc(&out, 4);
  struct { float acc0, b; int n; } cases[] = {
      {4194304.f, 1.0234375f, 1000},     // 2^22, b = 1 + 3/128 -> +32.75 per step
      {-4194304.f, 1.0234375f, 1000},    // negative accumulator: truncation toward zero vs toward -inf
      {4194304.f, -1.0234375f, 1000},
      {16384.f, 0.0009765625f * 1.0234375f, 1000},   // 32 products of ~2^-10: sum 0.03198, acc ulp at 2^14 = 2^-9 = 0.00195
      {0.f, 1.0234375f, 1000},
  };
  for (auto& c : cases) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, c.acc0, c.b, c.n);
    float h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
    double exact = (double)c.acc0 + 32.0 * (double)c.b * c.n;
    printf("acc0 %12.1f b %.10f n %d: got %.6f exact %.6f  per-step increment got %.6f exact %.6f\n", c.acc0, c.b, c.n, h, exact,
           (h - c.acc0) / c.n, 32.0 * c.b);
  }
  printf("32x32x16 (16 products per MFMA):\n");
  for (auto& c : cases) {
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, out, c.acc0, c.b, c.n);
    float h; hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost);
    printf("acc0 %12.1f b %.10f n %d: got %.6f  per-step increment got %.6f exact %.6f\n", c.acc0, c.b, c.n, h, (h - c.acc0) / c.n, 16.0 * c.b);
  }
  return 0;
}
