// Multi-head attention core *over the batch axis*.
//
// transformer_enc_layer feeds (B,S,E) tensors to nn.MultiheadAttention with batch_first=False
// (unet_cnns.py:134,153), so the sequence axis of the attention is the batch B and the "batch" is the S = T'*F'
// spatial positions.  For every position s and head h:
//     P = softmax_over_b'( d^-1/2 * q[b,s,h,:] . k[b',s,h,:] ),   o[b,s,h,:] = sum_b' P[b,b'] v[b',s,h,:]
// q,k,v,o are (B,S,E) row-major; head h owns columns [h*d, (h+1)*d).  d = E/heads <= 32.
// One workgroup per (s, h, 64 samples): lane = the wave's own sample b (a query in forward / backward-dq, a key in
// backward-dk,dv), the four waves split the *other* axis' samples four ways and merge through LDS, so that a batch of
// 256 gives 52*8*4 workgroups with 64 samples of serial work per thread instead of 416 workgroups with 256 -- the
// kernels are latency-bound VALU code (one dependent dot product + exp per pair) and need the extra waves per SIMD.
// Head dim d <= 32 is padded with zeros to the template width D, so the pair loops carry no predicates.
// < 0.05 % of the model's FLOPs.  Head dimension 16 (the paper's large configurations) runs on the matrix cores
// (attn16_* below); the other head dimensions keep these VALU kernels.
#include "mpa_common.h"
#include <algorithm>

namespace {

constexpr int DMAX = 32;
constexpr int KC = 128;      // samples of the other axis staged per LDS chunk (32 per wave)
constexpr int QB = 64;       // own samples per workgroup

template <int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ o,
                                                       float* __restrict__ lse, int B, int Bo, int S, int E, int heads,
                                                       float scale) {
  // B: queries (own samples, one per lane); Bo: keys / values (== B unless the keys of all data-parallel ranks were gathered)
  __shared__ float Ks[KC * D];
  __shared__ float Vs[KC * D];
  __shared__ float Ms[4][QB], Lsum[4][QB];
  const int s = blockIdx.x, h = blockIdx.y;
  const int d = E / heads;
  const long col0 = (long)s * E + h * d;
  const long rstride = (long)S * E;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.z * QB + lane;
  const bool active = b < B;
  float qr[D], acc[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    qr[j] = (active && j < d) ? q[(long)b * rstride + col0 + j] * scale : 0.f;
    acc[j] = 0.f;
  }
  float m = -INFINITY, l = 0.f;
  for (int c0 = 0; c0 < Bo; c0 += KC) {
    const int nk = min(KC, Bo - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < nk * D; e += 256) {
      const int kb = e / D, j = e - kb * D;
      const bool ok = j < d;
      Ks[e] = ok ? k[(long)(c0 + kb) * rstride + col0 + j] : 0.f;
      Vs[e] = ok ? v[(long)(c0 + kb) * rstride + col0 + j] : 0.f;
    }
    __syncthreads();
    const int kb_end = min(nk, (wave + 1) * (KC / 4));
    for (int kb = wave * (KC / 4); kb < kb_end; ++kb) {
      float sc = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) sc += qr[j] * Ks[kb * D + j];
      const float mn = fmaxf(m, sc);
      const float alpha = __expf(m - mn), pe = __expf(sc - mn);
      l = l * alpha + pe;
#pragma unroll
      for (int j = 0; j < D; ++j) acc[j] = acc[j] * alpha + pe * Vs[kb * D + j];
      m = mn;
    }
  }
  // merge the four waves' partial softmaxes of sample `lane`
  __syncthreads();
  Ms[wave][lane] = m;
  Lsum[wave][lane] = l;
  __syncthreads();
  float mstar = fmaxf(fmaxf(Ms[0][lane], Ms[1][lane]), fmaxf(Ms[2][lane], Ms[3][lane]));
  const float mine = (m == -INFINITY) ? 0.f : __expf(m - mstar);
  float* part = (wave < 2 ? Ks : Vs) + ((wave & 1) * QB + lane) * D;      // 2*QB*D = KC*D floats per array
#pragma unroll
  for (int j = 0; j < D; ++j) part[j] = acc[j] * mine;
  __syncthreads();
  if (wave == 0 && active) {
    float lstar = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) lstar += Ms[w][lane] == -INFINITY ? 0.f : Lsum[w][lane] * __expf(Ms[w][lane] - mstar);
    const float inv = 1.f / lstar;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float t = Ks[lane * D + j] + Ks[(QB + lane) * D + j] + Vs[lane * D + j] + Vs[(QB + lane) * D + j];
      if (j < d) o[(long)b * rstride + col0 + j] = t * inv;
    }
    lse[((long)s * heads + h) * B + b] = mstar + logf(lstar);
  }
}

// mode 0: lane = query b  -> dq[b]   = scale * sum_b' ds[b,b'] k[b']
// mode 1: lane = key  b'  -> dk[b']  = scale * sum_b  ds[b,b'] q[b],  dv[b'] = sum_b p[b,b'] do[b]
// ds = p * (do.v - D),  D[b] = do[b].o[b],  p = exp(scale*q.k - lse[b])
template <int MODE, int D>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ o,
                                                       const float* __restrict__ lse, const float* __restrict__ dO,
                                                       float* __restrict__ dq, float* __restrict__ dk,
                                                       float* __restrict__ dv, int B, int Bo, int BL, int S, int E, int heads,
                                                       float scale) {
  // B: own samples (queries in mode 0, keys in mode 1), Bo: the other axis, BL: number of queries (row length of lse)
  __shared__ float Xs[KC * D];      // mode 0: K chunk      mode 1: Q chunk
  __shared__ float Ys[KC * D];      // mode 0: V chunk      mode 1: dO chunk
  __shared__ float Ls[KC];          // mode 1: lse of the chunk's queries
  __shared__ float Ds[KC];          // mode 1: D of the chunk's queries
  const int s = blockIdx.x, h = blockIdx.y;
  const int d = E / heads;
  const long col0 = (long)s * E + h * d;
  const long rstride = (long)S * E;
  const float* lrow = lse + ((long)s * heads + h) * BL;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.z * QB + lane;
  const bool active = b < B;
  float r0[D], r1[D], a0[D], a1[D];
  float myl = 0.f, myD = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const bool ok = active && j < d;
    const long off = (long)b * rstride + col0 + j;
    if (MODE == 0) {
      r0[j] = ok ? q[off] : 0.f;
      r1[j] = ok ? dO[off] : 0.f;
      if (ok) myD += r1[j] * o[off];
    } else {
      r0[j] = ok ? k[off] : 0.f;
      r1[j] = ok ? v[off] : 0.f;
    }
    a0[j] = 0.f;
    a1[j] = 0.f;
  }
  if (MODE == 0 && active) myl = lrow[b];
  for (int c0 = 0; c0 < Bo; c0 += KC) {
    const int nk = min(KC, Bo - c0);
    __syncthreads();
    for (int e = threadIdx.x; e < nk * D; e += 256) {
      const int kb = e / D, j = e - kb * D;
      const long off = (long)(c0 + kb) * rstride + col0 + j;
      const bool ok = j < d;
      Xs[e] = ok ? (MODE == 0 ? k[off] : q[off]) : 0.f;
      Ys[e] = ok ? (MODE == 0 ? v[off] : dO[off]) : 0.f;
    }
    if (MODE == 1) {
      for (int kb = threadIdx.x; kb < nk; kb += 256) {
        Ls[kb] = lrow[c0 + kb];
        float dd = 0.f;
        for (int j = 0; j < d; ++j) {
          const long off = (long)(c0 + kb) * rstride + col0 + j;
          dd += dO[off] * o[off];
        }
        Ds[kb] = dd;
      }
    }
    __syncthreads();
    const int kb_end = min(nk, (wave + 1) * (KC / 4));
    for (int kb = wave * (KC / 4); kb < kb_end; ++kb) {
      float sc = 0.f, dp = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        sc += r0[j] * Xs[kb * D + j];
        dp += r1[j] * Ys[kb * D + j];
      }
      const float pr = __expf(sc * scale - (MODE == 0 ? myl : Ls[kb]));
      const float ds = pr * (dp - (MODE == 0 ? myD : Ds[kb])) * scale;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        a0[j] += ds * Xs[kb * D + j];
        if (MODE == 1) a1[j] += pr * Ys[kb * D + j];
      }
    }
  }
  // sum the four waves' partial results of sample `lane` (waves 0,1 -> Xs halves, 2,3 -> Ys halves; then a second
  // round for a1 in mode 1)
  for (int pass = 0; pass < (MODE == 1 ? 2 : 1); ++pass) {
    __syncthreads();
    float* part = (wave < 2 ? Xs : Ys) + ((wave & 1) * QB + lane) * D;
#pragma unroll
    for (int j = 0; j < D; ++j) part[j] = pass == 0 ? a0[j] : a1[j];
    __syncthreads();
    if (wave == 0 && active) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float t = Xs[lane * D + j] + Xs[(QB + lane) * D + j] + Ys[lane * D + j] + Ys[(QB + lane) * D + j];
        if (j < d) {
          const long off = (long)b * rstride + col0 + j;
          if (MODE == 0) dq[off] = t;
          else if (pass == 0) dk[off] = t;
          else dv[off] = t;
        }
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------ d = 16 on the matrix cores
// Head dimension 16 (E = 128, 8 heads: SAUnet:L / :XXL, SAUSnet:L, exp180d / exp181d) is exactly one group of four k steps of
// v_mfma_f32_16x16x4_f32, so both products of the attention run on the MFMA pipe in exact fp32:
//   S^T tile [16 keys][16 queries] = K_tile (A) x Q_tile^T (B)         -- a lane then holds S[query = l16][key = 4 kq + r]
//   O^T      [16 dd  ][16 queries] += V_tile^T (A) x P_tile^T (B)      -- B operand of k step r = the lane's own p[r]
// i.e. the probabilities go from the first product's accumulators (after exp) straight into the second product's B operand,
// no shuffles and no LDS round trip ("S^T trick"); the contraction index of every product is laid out as (kq, r) -> 4 kq + r so
// that a lane's share of an operand row is one 16-byte read.  The online-softmax maximum of a query is shared by its four kq
// lane groups (two ds_bpermute per key tile); the row sums stay lane-partial until the end.  A wave owns 16 queries (forward,
// dq) or 16 keys (dk, dv) and walks the other axis in LDS chunks of TK samples; operand images in LDS:
//   planes [kq][sample][4]   -- "row = sample" operands (A of S^T / dP^T): one ds_read_b128 per lane, the 16 lanes of a b128 group
//                               carry 16 distinct samples -> 16 distinct 16-byte slots, conflict-free;
//   rows   [sample][VP = 20] -- "row = dd" operands (A of the products that contract over samples): four ds_read_b32 per tile,
//                               samples 4 apart sit 16 banks apart -> conflict-free.
constexpr int TK = 128;
constexpr int VP = 20;

// n samples c0 .. c0+n-1 of a (B,S,E) tensor's 16-column head slice -> planes and / or rows (zeros past n)
__device__ __forceinline__ void attn16_stage(const float* __restrict__ src, long rstride, long col0, int c0, int n,
                                             float* __restrict__ planes, float* __restrict__ rows) {
  for (int e = threadIdx.x; e < TK * 4; e += 256) {
    const int row = e >> 2, kq = e & 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < n) v = *reinterpret_cast<const float4*>(src + (long)(c0 + row) * rstride + col0 + 4 * kq);
    if (planes) *reinterpret_cast<float4*>(planes + (kq * TK + row) * 4) = v;
    if (rows) *reinterpret_cast<float4*>(rows + row * VP + 4 * kq) = v;
  }
}

__device__ __forceinline__ f32x4 attn16_mma4(const float4 a, const float4 b, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
  return acc;
}

__global__ __launch_bounds__(256) void attn16_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, float* __restrict__ o,
                                                         float* __restrict__ lse, int Bq, int Bk, int S, int E, int heads,
                                                         float scale) {
  __shared__ __attribute__((aligned(16))) float Kpl[4 * TK * 4];
  __shared__ __attribute__((aligned(16))) float Vr[TK * VP];
  const int s = blockIdx.x, h = blockIdx.y;
  const long col0 = (long)s * E + h * 16, rstride = (long)S * E;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, l16 = lane & 15;
  const int qi = blockIdx.z * 64 + wave * 16 + l16;
  const bool qok = qi < Bq;
  float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (qok) {
    qv = *reinterpret_cast<const float4*>(q + (long)qi * rstride + col0 + 4 * kq);
    qv.x *= scale; qv.y *= scale; qv.z *= scale; qv.w *= scale;
  }
  f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  for (int c0 = 0; c0 < Bk; c0 += TK) {
    const int n = min(TK, Bk - c0);
    __syncthreads();
    attn16_stage(k, rstride, col0, c0, n, Kpl, nullptr);
    attn16_stage(v, rstride, col0, c0, n, nullptr, Vr);
    __syncthreads();
    const int ntile = (n + 15) >> 4;
    for (int kt = 0; kt < ntile; ++kt) {
      const float4 ka = *reinterpret_cast<const float4*>(Kpl + (kq * TK + kt * 16 + l16) * 4);
      f32x4 sa = attn16_mma4(ka, qv, f32x4{0.f, 0.f, 0.f, 0.f});
      const int key0 = c0 + kt * 16 + 4 * kq;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (key0 + r >= Bk) sa[r] = -INFINITY;
      float mx = fmaxf(fmaxf(sa[0], sa[1]), fmaxf(sa[2], sa[3]));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mn = fmaxf(m, mx);             // finite: every tile holds at least one key < Bk
      const float alpha = __expf(m - mn);
      float pr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[r] = __expf(sa[r] - mn);
      l = l * alpha + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc[r] *= alpha;
      m = mn;
      const float* vrow = Vr + (kt * 16 + 4 * kq) * VP + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) oacc = __builtin_amdgcn_mfma_f32_16x16x4f32(vrow[r * VP], pr[r], oacc, 0, 0, 0);
    }
  }
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  if (qok) {
    const float inv = 1.f / l;
    *reinterpret_cast<float4*>(o + (long)qi * rstride + col0 + 4 * kq) =
        make_float4(oacc[0] * inv, oacc[1] * inv, oacc[2] * inv, oacc[3] * inv);
    if (kq == 0) lse[((long)s * heads + h) * Bq + qi] = m + logf(l);
  }
}

// MODE 0: a wave owns 16 queries -> dq.  MODE 1: a wave owns 16 keys -> dk, dv (orientation "N": a lane holds
// S[query = 4 kq + r][key = l16], so that p and ds are the B operands of the products that contract over the queries).
template <int MODE>
__global__ __launch_bounds__(256) void attn16_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                         const float* __restrict__ v, const float* __restrict__ o,
                                                         const float* __restrict__ lse, const float* __restrict__ dO,
                                                         float* __restrict__ dq, float* __restrict__ dk, float* __restrict__ dv,
                                                         int Bq, int Bk, int S, int E, int heads, float scale) {
  __shared__ __attribute__((aligned(16))) float Apl[4 * TK * 4];   // MODE 0: K planes      MODE 1: Q planes
  __shared__ __attribute__((aligned(16))) float Ar[TK * VP];       // MODE 0: K rows        MODE 1: Q rows
  __shared__ __attribute__((aligned(16))) float Bpl[4 * TK * 4];   // MODE 0: V planes      MODE 1: dO planes
  __shared__ __attribute__((aligned(16))) float Br[MODE == 1 ? TK * VP : 4];          // MODE 1: dO rows
  __shared__ __attribute__((aligned(16))) float Ls[MODE == 1 ? TK : 4], Ds[MODE == 1 ? TK : 4];
  const int s = blockIdx.x, h = blockIdx.y;
  const long col0 = (long)s * E + h * 16, rstride = (long)S * E;
  const float* lrow = lse + ((long)s * heads + h) * Bq;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, l16 = lane & 15;
  const int own = blockIdx.z * 64 + wave * 16 + l16;           // this lane's query (MODE 0) / key (MODE 1)
  const int Bown = MODE == 0 ? Bq : Bk, Both = MODE == 0 ? Bk : Bq;
  const bool ok = own < Bown;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;        // MODE 0: q * scale, dO      MODE 1: k, v
  float myl = 0.f, myD = 0.f;
  if (ok) {
    const long off = (long)own * rstride + col0 + 4 * kq;
    if (MODE == 0) {
      r0 = *reinterpret_cast<const float4*>(q + off);
      r0.x *= scale; r0.y *= scale; r0.z *= scale; r0.w *= scale;
      r1 = *reinterpret_cast<const float4*>(dO + off);
      const float4 ov = *reinterpret_cast<const float4*>(o + off);
      myD = (r1.x * ov.x + r1.y * ov.y) + (r1.z * ov.z + r1.w * ov.w);
      myl = lrow[own];
    } else {
      r0 = *reinterpret_cast<const float4*>(k + off);
      r1 = *reinterpret_cast<const float4*>(v + off);
    }
  }
  if (MODE == 0) {
    myD += __shfl_xor(myD, 16, 64);
    myD += __shfl_xor(myD, 32, 64);
  }
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // MODE 0: dq^T      MODE 1: dk^T, dv^T
  for (int c0 = 0; c0 < Both; c0 += TK) {
    const int n = min(TK, Both - c0);
    __syncthreads();
    if (MODE == 0) {
      attn16_stage(k, rstride, col0, c0, n, Apl, Ar);
      attn16_stage(v, rstride, col0, c0, n, Bpl, nullptr);
    } else {
      attn16_stage(q, rstride, col0, c0, n, Apl, Ar);
      attn16_stage(dO, rstride, col0, c0, n, Bpl, Br);
      for (int kb = threadIdx.x; kb < TK; kb += 256) {
        float lv = INFINITY, dd = 0.f;                           // queries past Bq: p = exp(-inf) = 0
        if (kb < n) {
          lv = lrow[c0 + kb];
          const float* dp = dO + (long)(c0 + kb) * rstride + col0;
          const float* op = o + (long)(c0 + kb) * rstride + col0;
#pragma unroll
          for (int j = 0; j < 16; j += 4) {
            const float4 a = *reinterpret_cast<const float4*>(dp + j), b = *reinterpret_cast<const float4*>(op + j);
            dd += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
          }
        }
        Ls[kb] = lv;
        Ds[kb] = dd;
      }
    }
    __syncthreads();
    const int ntile = (n + 15) >> 4;
    for (int t = 0; t < ntile; ++t) {
      const float4 a0 = *reinterpret_cast<const float4*>(Apl + (kq * TK + t * 16 + l16) * 4);
      const float4 b0 = *reinterpret_cast<const float4*>(Bpl + (kq * TK + t * 16 + l16) * 4);
      const f32x4 sa = attn16_mma4(a0, r0, f32x4{0.f, 0.f, 0.f, 0.f});      // S^T (MODE 0, scaled) / S (MODE 1)
      const f32x4 dp = attn16_mma4(b0, r1, f32x4{0.f, 0.f, 0.f, 0.f});      // dP^T / dP
      float pr[4], ds[4];
      if (MODE == 0) {
        const int key0 = c0 + t * 16 + 4 * kq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = key0 + r < Bk ? __expf(sa[r] - myl) : 0.f;
          ds[r] = pr[r] * (dp[r] - myD) * scale;
        }
      } else {
        const float4 l4 = *reinterpret_cast<const float4*>(Ls + t * 16 + 4 * kq);
        const float4 d4 = *reinterpret_cast<const float4*>(Ds + t * 16 + 4 * kq);
        const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dv4[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = __expf(sa[r] * scale - lv[r]);
          ds[r] = pr[r] * (dp[r] - dv4[r]) * scale;
        }
      }
      const float* arow = Ar + (t * 16 + 4 * kq) * VP + l16;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[r * VP], ds[r], acc0, 0, 0, 0);
      if (MODE == 1) {
        const float* brow = Br + (t * 16 + 4 * kq) * VP + l16;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(brow[r * VP], pr[r], acc1, 0, 0, 0);
      }
    }
  }
  if (ok) {
    const long off = (long)own * rstride + col0 + 4 * kq;
    if (MODE == 0) {
      *reinterpret_cast<float4*>(dq + off) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
    } else {
      *reinterpret_cast<float4*>(dk + off) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
      *reinterpret_cast<float4*>(dv + off) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
    }
  }
}

// the MFMA kernels take head dimension 16 with 16-byte-aligned head slices
inline bool attn16_ok(int E, int heads, const void* a, const void* b, const void* c, const void* d2) {
  return !mpa_diag().attn_valu && E / heads == 16 && E % 4 == 0 &&
         ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
           reinterpret_cast<uintptr_t>(d2)) & 15) == 0;
}

template <int D>
void launch_attn_fwd(dim3 grid, hipStream_t st, const float* q, const float* k, const float* v, float* o, float* lse, int Bq,
                     int Bk, int S, int E, int heads, float scale) {
  MPA_LAUNCH((attn_fwd_kernel<D>), grid, dim3(256), 0, st, q, k, v, o, lse, Bq, Bk, S, E, heads, scale);
}
template <int D>
void launch_attn_bwd(dim3 grid, hipStream_t st, const float* q, const float* k, const float* v, const float* o,
                     const float* lse, const float* dO, float* dq, float* dk, float* dv, int Bq, int Bk, int S, int E,
                     int heads, float scale) {
  grid.z = (unsigned)mpa_cdiv(Bq, QB);
  MPA_LAUNCH((attn_bwd_kernel<0, D>), grid, dim3(256), 0, st, q, k, v, o, lse, dO, dq, dk, dv, Bq, Bk, Bq, S, E, heads, scale);
  grid.z = (unsigned)mpa_cdiv(Bk, QB);
  MPA_LAUNCH((attn_bwd_kernel<1, D>), grid, dim3(256), 0, st, q, k, v, o, lse, dO, dq, dk, dv, Bk, Bq, Bq, S, E, heads, scale);
}

}  // namespace

extern "C" {

int mpa_attn_batchaxis_fwd_kv(const float* q, const float* k, const float* v, float* o, float* lse, int Bq, int Bk, int S,
                              int E, int heads, void* stream) {
  if (!q || !k || !v || !o || !lse || heads <= 0 || E % heads || E / heads > DMAX || Bq <= 0 || Bk <= 0) return MPA_ERR_ARG;
  const float scale = 1.0f / sqrtf((float)(E / heads));
  const int d = E / heads;
  const dim3 grid(S, heads, (unsigned)mpa_cdiv(Bq, QB));
  hipStream_t st = (hipStream_t)stream;
  if (attn16_ok(E, heads, q, k, v, o)) {
    MPA_LAUNCH(attn16_fwd_kernel, grid, dim3(256), 0, st, q, k, v, o, lse, Bq, Bk, S, E, heads, scale);
    return mpa_launch_status();
  }
  if (d <= 4) launch_attn_fwd<4>(grid, st, q, k, v, o, lse, Bq, Bk, S, E, heads, scale);
  else if (d <= 8) launch_attn_fwd<8>(grid, st, q, k, v, o, lse, Bq, Bk, S, E, heads, scale);
  else if (d <= 16) launch_attn_fwd<16>(grid, st, q, k, v, o, lse, Bq, Bk, S, E, heads, scale);
  else launch_attn_fwd<32>(grid, st, q, k, v, o, lse, Bq, Bk, S, E, heads, scale);
  return mpa_launch_status();
}

int mpa_attn_batchaxis_bwd_kv(const float* q, const float* k, const float* v, const float* o, const float* lse,
                              const float* do_, float* dq, float* dk, float* dv, int Bq, int Bk, int S, int E, int heads,
                              void* stream) {
  if (!q || !k || !v || !o || !lse || !do_ || !dq || !dk || !dv || heads <= 0 || E % heads || E / heads > DMAX || Bq <= 0 ||
      Bk <= 0)
    return MPA_ERR_ARG;
  const float scale = 1.0f / sqrtf((float)(E / heads));
  hipStream_t s = (hipStream_t)stream;
  const int d = E / heads;
  if (attn16_ok(E, heads, q, k, v, o) && attn16_ok(E, heads, do_, dq, dk, dv)) {
    MPA_LAUNCH(attn16_bwd_kernel<0>, dim3(S, heads, (unsigned)mpa_cdiv(Bq, 64)), dim3(256), 0, s, q, k, v, o, lse, do_, dq, dk, dv,
               Bq, Bk, S, E, heads, scale);
    MPA_LAUNCH(attn16_bwd_kernel<1>, dim3(S, heads, (unsigned)mpa_cdiv(Bk, 64)), dim3(256), 0, s, q, k, v, o, lse, do_, dq, dk, dv,
               Bq, Bk, S, E, heads, scale);
    return mpa_launch_status();
  }
  const dim3 grid(S, heads, 1);
  if (d <= 4) launch_attn_bwd<4>(grid, s, q, k, v, o, lse, do_, dq, dk, dv, Bq, Bk, S, E, heads, scale);
  else if (d <= 8) launch_attn_bwd<8>(grid, s, q, k, v, o, lse, do_, dq, dk, dv, Bq, Bk, S, E, heads, scale);
  else if (d <= 16) launch_attn_bwd<16>(grid, s, q, k, v, o, lse, do_, dq, dk, dv, Bq, Bk, S, E, heads, scale);
  else launch_attn_bwd<32>(grid, s, q, k, v, o, lse, do_, dq, dk, dv, Bq, Bk, S, E, heads, scale);
  return mpa_launch_status();
}

int mpa_attn_batchaxis_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int S, int E,
                           int heads, void* stream) {
  return mpa_attn_batchaxis_fwd_kv(q, k, v, o, lse, B, B, S, E, heads, stream);
}

int mpa_attn_batchaxis_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse,
                           const float* do_, float* dq, float* dk, float* dv, int B, int S, int E, int heads, void* stream) {
  return mpa_attn_batchaxis_bwd_kv(q, k, v, o, lse, do_, dq, dk, dv, B, B, S, E, heads, stream);
}

}  // extern "C"
