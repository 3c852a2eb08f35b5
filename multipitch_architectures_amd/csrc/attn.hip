// Multi-head attention core *over the batch axis*.
//
// transformer_enc_layer feeds (B,S,E) tensors to nn.MultiheadAttention with batch_first=False
// (unet_cnns.py:134,153), so the sequence axis of the attention is the batch B and the "batch" is the S = T'*F'
// spatial positions.  For every position s and head h:
//     P = softmax_over_b'( d^-1/2 * q[b,s,h,:] . k[b',s,h,:] ),   o[b,s,h,:] = sum_b' P[b,b'] v[b',s,h,:]
// q,k,v,o are (B,S,E) row-major; head h owns columns [h*d, (h+1)*d).  d = E/heads <= 32.
// One block per (s,h); one thread per query sample b (looped if B > blockDim); keys/values staged in LDS in
// chunks of KC samples with an online softmax.  < 0.05 % of the model's FLOPs, so plain VALU fp32.
#include "mpa_common.h"
#include <algorithm>

namespace {

constexpr int DMAX = 32;
constexpr int KC = 128;

__global__ __launch_bounds__(256) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ o,
                                                       float* __restrict__ lse, int B, int S, int E, int heads, float scale) {
  __shared__ float Ks[KC * DMAX];
  __shared__ float Vs[KC * DMAX];
  const int s = blockIdx.x, h = blockIdx.y;
  const int d = E / heads;
  const long col0 = (long)s * E + h * d;
  const long rstride = (long)S * E;
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const bool active = b < B;
    float qr[DMAX], acc[DMAX];
#pragma unroll
    for (int j = 0; j < DMAX; ++j) {
      qr[j] = (active && j < d) ? q[(long)b * rstride + col0 + j] * scale : 0.f;
      acc[j] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int c0 = 0; c0 < B; c0 += KC) {
      const int nk = min(KC, B - c0);
      __syncthreads();
      for (int e = threadIdx.x; e < nk * d; e += 256) {
        const int kb = e / d, j = e - kb * d;
        Ks[kb * DMAX + j] = k[(long)(c0 + kb) * rstride + col0 + j];
        Vs[kb * DMAX + j] = v[(long)(c0 + kb) * rstride + col0 + j];
      }
      __syncthreads();
      if (active) {
        for (int kb = 0; kb < nk; ++kb) {
          float sc = 0.f;
#pragma unroll
          for (int j = 0; j < DMAX; ++j)
            if (j < d) sc += qr[j] * Ks[kb * DMAX + j];
          const float mn = fmaxf(m, sc);
          const float alpha = expf(m - mn), pe = expf(sc - mn);
          l = l * alpha + pe;
#pragma unroll
          for (int j = 0; j < DMAX; ++j)
            if (j < d) acc[j] = acc[j] * alpha + pe * Vs[kb * DMAX + j];
          m = mn;
        }
      }
    }
    if (active) {
      const float inv = 1.f / l;
#pragma unroll
      for (int j = 0; j < DMAX; ++j)
        if (j < d) o[(long)b * rstride + col0 + j] = acc[j] * inv;
      lse[((long)s * heads + h) * B + b] = m + logf(l);
    }
  }
}

// mode 0: thread = query b  -> dq[b]   = scale * sum_b' ds[b,b'] k[b']
// mode 1: thread = key  b'  -> dk[b']  = scale * sum_b  ds[b,b'] q[b],  dv[b'] = sum_b p[b,b'] do[b]
// ds = p * (do.v - D),  D[b] = do[b].o[b],  p = exp(scale*q.k - lse[b])
template <int MODE>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ o,
                                                       const float* __restrict__ lse, const float* __restrict__ dO,
                                                       float* __restrict__ dq, float* __restrict__ dk,
                                                       float* __restrict__ dv, int B, int S, int E, int heads, float scale) {
  __shared__ float Xs[KC * DMAX];   // mode 0: K chunk      mode 1: Q chunk
  __shared__ float Ys[KC * DMAX];   // mode 0: V chunk      mode 1: dO chunk
  __shared__ float Ls[KC];          // mode 1: lse of the chunk's queries
  __shared__ float Ds[KC];          // mode 1: D of the chunk's queries
  const int s = blockIdx.x, h = blockIdx.y;
  const int d = E / heads;
  const long col0 = (long)s * E + h * d;
  const long rstride = (long)S * E;
  const float* lrow = lse + ((long)s * heads + h) * B;
  for (int b0 = 0; b0 < B; b0 += 256) {
    const int b = b0 + threadIdx.x;
    const bool active = b < B;
    float r0[DMAX], r1[DMAX], a0[DMAX], a1[DMAX];
    float myl = 0.f, myD = 0.f;
#pragma unroll
    for (int j = 0; j < DMAX; ++j) {
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
    for (int c0 = 0; c0 < B; c0 += KC) {
      const int nk = min(KC, B - c0);
      __syncthreads();
      for (int e = threadIdx.x; e < nk * d; e += 256) {
        const int kb = e / d, j = e - kb * d;
        const long off = (long)(c0 + kb) * rstride + col0 + j;
        Xs[kb * DMAX + j] = MODE == 0 ? k[off] : q[off];
        Ys[kb * DMAX + j] = MODE == 0 ? v[off] : dO[off];
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
      if (active) {
        for (int kb = 0; kb < nk; ++kb) {
          float sc = 0.f, dp = 0.f;
#pragma unroll
          for (int j = 0; j < DMAX; ++j)
            if (j < d) {
              sc += r0[j] * Xs[kb * DMAX + j];
              dp += (MODE == 0 ? r1[j] * Ys[kb * DMAX + j] : Ys[kb * DMAX + j] * r1[j]);
            }
          const float pr = expf(sc * scale - (MODE == 0 ? myl : Ls[kb]));
          const float ds = pr * (dp - (MODE == 0 ? myD : Ds[kb])) * scale;
#pragma unroll
          for (int j = 0; j < DMAX; ++j)
            if (j < d) {
              a0[j] += ds * Xs[kb * DMAX + j];
              if (MODE == 1) a1[j] += pr * Ys[kb * DMAX + j];
            }
        }
      }
    }
    if (active) {
#pragma unroll
      for (int j = 0; j < DMAX; ++j)
        if (j < d) {
          const long off = (long)b * rstride + col0 + j;
          if (MODE == 0) dq[off] = a0[j];
          else { dk[off] = a0[j]; dv[off] = a1[j]; }
        }
    }
  }
}

}  // namespace

extern "C" {

int mpa_attn_batchaxis_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int S, int E,
                           int heads, void* stream) {
  if (!q || !k || !v || !o || !lse || heads <= 0 || E % heads || E / heads > DMAX) return MPA_ERR_ARG;
  const float scale = 1.0f / sqrtf((float)(E / heads));
  MPA_LAUNCH(attn_fwd_kernel, dim3(S, heads), dim3(256), 0, (hipStream_t)stream, q, k, v, o, lse, B, S, E, heads,
                     scale);
  return mpa_launch_status();
}

int mpa_attn_batchaxis_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse,
                           const float* do_, float* dq, float* dk, float* dv, int B, int S, int E, int heads, void* stream) {
  if (!q || !k || !v || !o || !lse || !do_ || !dq || !dk || !dv || heads <= 0 || E % heads || E / heads > DMAX)
    return MPA_ERR_ARG;
  const float scale = 1.0f / sqrtf((float)(E / heads));
  hipStream_t s = (hipStream_t)stream;
  MPA_LAUNCH((attn_bwd_kernel<0>), dim3(S, heads), dim3(256), 0, s, q, k, v, o, lse, do_, dq, dk, dv, B, S, E, heads,
                     scale);
  MPA_LAUNCH((attn_bwd_kernel<1>), dim3(S, heads), dim3(256), 0, s, q, k, v, o, lse, do_, dq, dk, dv, B, S, E, heads,
                     scale);
  return mpa_launch_status();
}

}  // extern "C"
