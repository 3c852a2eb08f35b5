// does global_load_lds_dwordx4 accept a global address that is only 4-byte aligned?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* src, float* out, int shift) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4];
  const int lane = threadIdx.x;
  // each lane copies 4 floats starting at src[shift + 4*lane] into lds[4*lane..]
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + shift + 4 * lane),
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}
int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, 4096); hipMalloc(&o, 1024);
  hipMemcpy(d, h.data(), 4096, hipMemcpyHostToDevice);
  for (int shift = 0; shift < 8; ++shift) {
    hipMemset(o, 0, 1024);
    k<<<1, 64>>>(d, o, shift);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (r[i] != (float)(i + shift)) ++bad;
    printf("shift %d: %d/256 wrong (%s) first: %g %g %g %g %g\n", shift, bad, hipGetErrorString(e), r[0], r[1], r[2], r[3], r[4]);
  }
  return 0;
}
