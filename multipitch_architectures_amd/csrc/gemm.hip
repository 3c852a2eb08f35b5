// Strided fp32 GEMM on v_mfma_f32_16x16x4_f32 for nn.Linear (unet_cnns.py:131-141) and nn.LSTM
// projections (unet_cnns.py:232):  C[M,N] (+)= A[M,K] * B[K,N] (+ bias[N]) -> act
//   A(m,k) = A[m*lda_m + k*lda_k],  B(k,n) = Bm[k*ldb_k + n*ldb_n]   (any of NN / NT / TN / TT)
// Block tiles of 128x128 / 128x64 / 64x128 / 64x64 (picked so the grid still fills 256 CUs), BK = 32, 2x2 waves.
// The LDS image of each operand is laid out along whichever global dimension is contiguous so that both the
// global loads and the LDS stores stay coalesced / conflict-free; MFMA fragment reads use runtime strides.
#include "mpa_common.h"
#include <algorithm>

namespace {

constexpr int BK = 32;

struct GemmParams {
  const float* A;
  const float* B;
  const float* bias;
  float* C;
  long lda_m, lda_k, ldb_k, ldb_n, ldc;
  int M, N, K, accumulate, act;
};

// stage an (R x BK) operand tile: elem(r,k) = src[r*ld_r + k*ld_k]; k contiguous -> r-major image (pitch BK+2),
// else k-major image (pitch R+16)
template <int R>
__device__ __forceinline__ void stage_operand(float* __restrict__ img, const float* __restrict__ src, long ld_r, long ld_k,
                                              int r0, int k0, int Rlim, int K, int tid) {
  float v[(R * BK) / 256];
  if (ld_k == 1) {
#pragma unroll
    for (int it = 0; it < (R * BK) / 256; ++it) {
      const int e = it * 256 + tid;
      const int r = e / BK, k = e % BK;
      v[it] = (r0 + r < Rlim && k0 + k < K) ? src[(long)(r0 + r) * ld_r + (k0 + k)] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < (R * BK) / 256; ++it) {
      const int e = it * 256 + tid;
      img[(e / BK) * (BK + 2) + (e % BK)] = v[it];
    }
  } else {
#pragma unroll
    for (int it = 0; it < (R * BK) / 256; ++it) {
      const int e = it * 256 + tid;
      const int k = e / R, r = e % R;
      v[it] = (r0 + r < Rlim && k0 + k < K) ? src[(long)(r0 + r) * ld_r + (long)(k0 + k) * ld_k] : 0.f;
    }
#pragma unroll
    for (int it = 0; it < (R * BK) / 256; ++it) {
      const int e = it * 256 + tid;
      img[(e / R) * (R + 16) + (e % R)] = v[it];
    }
  }
}

// block = 2x2 waves, wave tile = (WM*16) x (WN*16)
template <int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int BMt = 2 * WM * 16, BNt = 2 * WN * 16;
  constexpr int AF = (BMt * (BK + 2) > BK * (BMt + 16)) ? BMt * (BK + 2) : BK * (BMt + 16);
  constexpr int BF = (BNt * (BK + 2) > BK * (BNt + 16)) ? BNt * (BK + 2) : BK * (BNt + 16);
  __shared__ __attribute__((aligned(16))) float As[AF];
  __shared__ __attribute__((aligned(16))) float Bs[BF];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * BMt, n0 = blockIdx.x * BNt;
  const int wm = (wave >> 1) * WM * 16, wn = (wave & 1) * WN * 16;
  const int kq = lane >> 4, l16 = lane & 15;
  const int a_sr = p.lda_k == 1 ? (BK + 2) : 1, a_sk = p.lda_k == 1 ? 1 : (BMt + 16);
  const int b_sr = p.ldb_k == 1 ? (BK + 2) : 1, b_sk = p.ldb_k == 1 ? 1 : (BNt + 16);
  f32x4 acc[WM][WN];
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < p.K; k0 += BK) {
    __syncthreads();
    stage_operand<BMt>(As, p.A, p.lda_m, p.lda_k, m0, k0, p.M, p.K, tid);
    stage_operand<BNt>(Bs, p.B, p.ldb_n, p.ldb_k, n0, k0, p.N, p.K, tid);
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; kk += 4) {
      float a[WM], b[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) a[i] = As[(wm + i * 16 + l16) * a_sr + (kk + kq) * a_sk];
#pragma unroll
      for (int j = 0; j < WN; ++j) b[j] = Bs[(wn + j * 16 + l16) * b_sr + (kk + kq) * b_sk];
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
  // D[m = kq*4 + r][n = l16]
#pragma unroll
  for (int i = 0; i < WM; ++i)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
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

template <int WM, int WN>
void launch_gemm(const GemmParams& p, hipStream_t s) {
  dim3 grid((unsigned)mpa_cdiv(p.N, 2 * WN * 16), (unsigned)mpa_cdiv(p.M, 2 * WM * 16));
  MPA_LAUNCH((gemm_kernel<WM, WN>), grid, dim3(256), 0, s, p);
}

}  // namespace

extern "C" int mpa_gemm(const float* A, int64_t lda_m, int64_t lda_k, const float* Bm, int64_t ldb_k, int64_t ldb_n,
                        const float* bias, float* C, int64_t ldc, int M, int N, int K, int accumulate, int act,
                        void* stream) {
  if (!A || !Bm || !C || M <= 0 || N <= 0 || K <= 0) return MPA_ERR_ARG;
  GemmParams p{A, Bm, bias, C, (long)lda_m, (long)lda_k, (long)ldb_k, (long)ldb_n, (long)ldc, M, N, K, accumulate, act};
  hipStream_t s = (hipStream_t)stream;
  // largest tile that still gives every CU (256) a couple of workgroups
  const long b128 = mpa_cdiv(M, 128) * mpa_cdiv(N, 128);
  const long b12864 = mpa_cdiv(M, 128) * mpa_cdiv(N, 64);
  const long b64128 = mpa_cdiv(M, 64) * mpa_cdiv(N, 128);
  if (b128 >= 512) launch_gemm<4, 4>(p, s);
  else if (b12864 >= 384 && M >= N) launch_gemm<4, 2>(p, s);
  else if (b64128 >= 384) launch_gemm<2, 4>(p, s);
  else if (b12864 >= 384) launch_gemm<4, 2>(p, s);
  else launch_gemm<2, 2>(p, s);
  return mpa_launch_status();
}
