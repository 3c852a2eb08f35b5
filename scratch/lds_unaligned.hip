// does ds_read_b128 / ds_read_b64 work at 4-byte alignment on gfx950, and how fast is it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int shift, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  unsigned addr = (unsigned)((lane + shift) * 4 + (threadIdx.x >> 6) * 4096);   // byte address: lane-consecutive floats
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {          // 4 x b32 via 2 x read2
      float a, b, c, d;
      asm volatile("ds_read2_b32 %0, %2 offset1:1\n ds_read2_b32 %1, %2 offset0:2 offset1:3\n s_waitcnt lgkmcnt(0)"
                   : "=v"(*(f32x2*)&a), "=v"(*(f32x2*)&c) : "v"(addr) : "memory");
      acc += a;
    } else if (MODE == 1) {   // b128
      f32x4 v;
      asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
      acc += v[0] + v[1] + v[2] + v[3];
    } else {                  // b64 x2
      f32x2 v, w;
      asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(v), "=v"(w) : "v"(addr) : "memory");
      acc += v[0] + v[1] + w[0] + w[1];
    }
    addr ^= 64;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
void run(const char* name, int shift) {
  float* d; hipMalloc(&d, 256 * 1024 * 4);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<1024, 256>>>(d, shift, 10);
  hipEventRecord(e0);
  k<MODE><<<1024, 256>>>(d, shift, iters);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // correctness of one read
  k<MODE><<<1, 256>>>(d, shift, 1);
  std::vector<float> h(256); hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int t = 0; t < 256; ++t) {
    int base = (t & 63) + shift + (t >> 6) * 1024;
    float want = MODE == 0 ? (float)base : (float)(4 * base + 6);
    if (h[t] != want) ++bad;
  }
  printf("%s shift=%d: %.3f ms, %d/256 wrong, hipError=%s\n", name, shift, ms, bad, hipGetErrorString(hipGetLastError()));
  hipFree(d);
}
int main() {
  for (int shift = 0; shift < 4; ++shift) {
    run<0>("2x read2_b32", shift);
    run<1>("read_b128   ", shift);
    run<2>("2x read_b64 ", shift);
  }
  return 0;
}
