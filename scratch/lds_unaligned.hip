// does ds_read_b128 / ds_read_b64 work at 4-byte alignment on gfx950, and how fast is it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef LSTRIDE
#define LSTRIDE 4
#endif
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int shift, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192 + 2048];
  for (int i = threadIdx.x; i < 8192 + 2048; i += 256) lds[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  unsigned addr = (unsigned)((lane * LSTRIDE + shift) * 4 + (threadIdx.x >> 6) * 8192);   // byte address: lanes LSTRIDE floats apart
  for (int it = 0; it < iters; ++it) {
    // 4 independent requests of 4 floats each in flight, one wait
    if (MODE == 0) {
      f32x2 r[8];
      asm volatile("ds_read2_b32 %0, %8 offset1:1\n ds_read2_b32 %1, %8 offset0:2 offset1:3\n"
                   "ds_read2_b32 %2, %8 offset0:16 offset1:17\n ds_read2_b32 %3, %8 offset0:18 offset1:19\n"
                   "ds_read2_b32 %4, %8 offset0:32 offset1:33\n ds_read2_b32 %5, %8 offset0:34 offset1:35\n"
                   "ds_read2_b32 %6, %8 offset0:48 offset1:49\n ds_read2_b32 %7, %8 offset0:50 offset1:51\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
                   : "v"(addr) : "memory");
      acc += r[0][0] + r[0][1] + r[1][0] + r[1][1] + r[2][0] + r[4][0] + r[6][0];
    } else if (MODE == 1) {
      f32x4 v[4];
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:64\n ds_read_b128 %2, %4 offset:128\n"
                   "ds_read_b128 %3, %4 offset:192\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr) : "memory");
      acc += v[0][0] + v[0][1] + v[0][2] + v[0][3] + v[1][0] + v[2][0] + v[3][0];
    } else {
      f32x2 v[8];
      asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:64\n ds_read_b64 %3, %8 offset:72\n"
                   "ds_read_b64 %4, %8 offset:128\n ds_read_b64 %5, %8 offset:136\n ds_read_b64 %6, %8 offset:192\n"
                   "ds_read_b64 %7, %8 offset:200\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                   : "v"(addr) : "memory");
      acc += v[0][0] + v[0][1] + v[1][0] + v[1][1] + v[2][0] + v[4][0] + v[6][0];
    }
    addr ^= 2048;
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
    int base = (t & 63) * LSTRIDE + shift + (t >> 6) * 2048;
    float want = (float)(4 * base + 6 + 3 * base + 16 + 32 + 48);
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
