// probe: what do HW_REG_LDS_ALLOC / HW_REG_HW_ID report for co-resident workgroups? (diagnostic, not product code)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
__global__ __launch_bounds__(256) void probe(unsigned* out) {
  extern __shared__ float lds[];
  unsigned a, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(a));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  lds[threadIdx.x] = 1.f;
  __syncthreads();
  // keep the workgroup alive for a while so that two are co-resident
  float s = 0.f;
  for (int i = 0; i < 20000; ++i) s += lds[(threadIdx.x + i) & 255];
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = a; out[blockIdx.x * 2 + 1] = h + (s < 0.f ? 1 : 0); }
}
int main() {
  const int n = 1024;
  unsigned* d; hipMalloc(&d, n * 8);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 76288, 0, d);
  unsigned h[2 * n]; hipMemcpy(h, d, n * 8, hipMemcpyDeviceToHost);
  std::map<unsigned, int> ha, tg, wv;
  for (int i = 0; i < n; ++i) { ha[h[2 * i]]++; tg[(h[2 * i + 1] >> 16) & 15]++; wv[h[2 * i + 1] & 15]++; }
  printf("LDS_ALLOC values:\n"); for (auto& kv : ha) printf("  %08x base=%u size=%u : %d\n", kv.first, kv.first & 255, (kv.first >> 12) & 511, kv.second);
  printf("TG_ID:"); for (auto& kv : tg) printf(" %u:%d", kv.first, kv.second); printf("\nWAVE_ID:"); for (auto& kv : wv) printf(" %u:%d", kv.first, kv.second); printf("\n");
  for (int i = 0; i < 8; ++i) printf("wg %d: alloc %08x hwid %08x\n", i, h[2 * i], h[2 * i + 1]);
  return 0;
}
