// Strided fp32 GEMM on v_mfma_f32_16x16x4_f32 for nn.Linear (unet_cnns.py:131-141) and nn.LSTM
// projections (unet_cnns.py:232):  C[M,N] (+)= A[M,K] * B[K,N] (+ bias[N]) -> act
//   A(m,k) = A[m*lda_m + k*lda_k],  B(k,n) = Bm[k*ldb_k + n*ldb_n]   (any of NN / NT / TN / TT)
// 64x64 block tile, BK = 32, 4 waves each owning a 32x32 sub-tile (2x2 MFMA blocks).
// The LDS image of each operand is laid out along whichever global dimension is contiguous so that both the
// global loads and the LDS stores stay coalesced / conflict-free; MFMA fragment reads use runtime strides.
#include "mpa_common.h"
#include <algorithm>

namespace {

constexpr int BM = 64, BN = 64, BK = 32;
constexpr int TILE_F = 64 * 34 > 32 * 80 ? 64 * 34 : 32 * 80;   // floats per operand image

struct GemmParams {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  long lda_m, lda_k, ldb_k, ldb_n, ldc;
  int M, N, K, accumulate, act;
};

// stage a (R x BK) operand tile: elem(r,k) = src[r*ld_r + k*ld_k]; k contiguous -> r-major image (pitch BK+2),
// else k-major image (pitch 64+16)
__device__ __forceinline__ void stage_operand(float* __restrict__ img, const float* __restrict__ src, long ld_r, long ld_k,
                                              int r0, int k0, int R, int K, int tid) {
  if (ld_k == 1) {
#pragma unroll
    for (int it = 0; it < (64 * BK) / 256; ++it) {
      const int e = it * 256 + tid;
      const int r = e / BK, k = e % BK;
      float v = 0.f;
      if (r0 + r < R && k0 + k < K) v = src[(long)(r0 + r) * ld_r + (k0 + k)];
      img[r * (BK + 2) + k] = v;
    }
  } else {
#pragma unroll
    for (int it = 0; it < (64 * BK) / 256; ++it) {
      const int e = it * 256 + tid;
      const int k = e / 64, r = e % 64;
      float v = 0.f;
      if (r0 + r < R && k0 + k < K) v = src[(long)(r0 + r) * ld_r + (long)(k0 + k) * ld_k];
      img[k * 80 + r] = v;
    }
  }
}

__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  __shared__ __attribute__((aligned(16))) float As[TILE_F];
  __shared__ __attribute__((aligned(16))) float Bs[TILE_F];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
  const int kq = lane >> 4, l16 = lane & 15;
  const int a_sr = p.lda_k == 1 ? (BK + 2) : 1, a_sk = p.lda_k == 1 ? 1 : 80;
  const int b_sr = p.ldb_k == 1 ? (BK + 2) : 1, b_sk = p.ldb_k == 1 ? 1 : 80;
  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < p.K; k0 += BK) {
    __syncthreads();
    stage_operand(As, p.A, p.lda_m, p.lda_k, m0, k0, p.M, p.K, tid);
    stage_operand(Bs, p.B, p.ldb_n, p.ldb_k, n0, k0, p.N, p.K, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = As[(wm + i * 16 + l16) * a_sr + (kk + kq) * a_sk];
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = Bs[(wn + j * 16 + l16) * b_sr + (kk + kq) * b_sk];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // D[m = kq*4 + r][n = l16]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wn + j * 16 + l16;
      if (n >= p.N) continue;
      const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm + i * 16 + kq * 4 + r;
        if (m >= p.M) continue;
        float v = acc[i][j][r] + bv;
        float* c = p.C + (long)m * p.ldc + n;
        if (p.accumulate) v += *c;
        *c = mpa_apply_act(v, p.act, 0.f);
      }
    }
}

}  // namespace

extern "C" int mpa_gemm(const float* A, int64_t lda_m, int64_t lda_k, const float* Bm, int64_t ldb_k, int64_t ldb_n,
                        const float* bias, float* C, int64_t ldc, int M, int N, int K, int accumulate, int act,
                        void* stream) {
  if (!A || !Bm || !C || M <= 0 || N <= 0 || K <= 0) return MPA_ERR_ARG;
  GemmParams p{A, Bm, bias, C, (long)lda_m, (long)lda_k, (long)ldb_k, (long)ldb_n, (long)ldc, M, N, K, accumulate, act};
  dim3 grid((unsigned)mpa_cdiv(N, BN), (unsigned)mpa_cdiv(M, BM));
  hipLaunchKernelGGL(gemm_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  return mpa_launch_status();
}
