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
// < 0.05 % of the model's FLOPs, so plain VALU fp32.
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
