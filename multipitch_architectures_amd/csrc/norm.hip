// Normalisation kernels (HBM-bound): input LayerNorm([C,F]), BatchNorm2d(+ReLU), LayerNorm over rows.
#include "mpa_common.h"
#include <algorithm>

namespace {

// ---------------------------------------------------------------------------------- LayerNorm([C,F]) per (b,t)
// x (B,C,T,F).  One wave per (b,t): C rows of F contiguous floats.  Two-pass statistics held in registers.
constexpr int LNCF_MAXV = 24;   // C*F <= 64*24 = 1536 values per (b,t)

__global__ __launch_bounds__(256) void layernorm_cf_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bb, float* __restrict__ y,
                                                               float* __restrict__ mean, float* __restrict__ rstd, int B,
                                                               int C, int T, int F, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long bt = (long)blockIdx.x * 4 + wave;
  if (bt >= (long)B * T) return;
  const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
  const int n = C * F;
  const long rowbase = (long)b * C * T * F + (long)t * F;      // element (c, f) of the row sits at rowbase + c T F + f
  // every lane loads all LNCF_MAXV slots unconditionally (slots past n re-read element 0 and are masked afterwards):
  // guarded loads compile to one branch + wait per element, i.e. LNCF_MAXV serial trips to memory per row
  float v[LNCF_MAXV];
  int off[LNCF_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LNCF_MAXV; ++i) {
    const int e = min(i * 64 + lane, n - 1);
    const int c = e / F, f = e - c * F;
    off[i] = c * T * F + f;
  }
#pragma unroll
  for (int i = 0; i < LNCF_MAXV; ++i) v[i] = x[rowbase + off[i]];
#pragma unroll
  for (int i = 0; i < LNCF_MAXV; ++i) {
    v[i] = i * 64 + lane < n ? v[i] : 0.f;
    s += v[i];
  }
  const float mu = mpa_wave_sum(s) / (float)n;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LNCF_MAXV; ++i) {
    const int e = i * 64 + lane;
    if (e < n) { const float d = v[i] - mu; q += d * d; }
  }
  const float rs = 1.0f / sqrtf(mpa_wave_sum(q) / (float)n + eps);
#pragma unroll
  for (int i = 0; i < LNCF_MAXV; ++i) {
    const int e = i * 64 + lane;
    if (e < n) y[rowbase + off[i]] = (v[i] - mu) * rs * w[e] + bb[e];      // (w, bb: L1 hits)
  }
  if (lane == 0) { mean[bt] = mu; rstd[bt] = rs; }
}

// backward: per-block partial dw/db over a strided set of (b,t) rows -> ws[blk][2][n]; optional dx.
__global__ __launch_bounds__(256) void layernorm_cf_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ w, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, float* __restrict__ dx,
                                                               float* __restrict__ ws, int B, int C, int T, int F) {
  extern __shared__ float sh[];   // [4][2][n] per-wave partials
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = C * F;
  float pw[LNCF_MAXV], pb[LNCF_MAXV], wv[LNCF_MAXV];
  int off[LNCF_MAXV];      // element (c, f) of a row sits at rowbase + c T F + f: the division by F once per kernel
#pragma unroll
  for (int i = 0; i < LNCF_MAXV; ++i) {
    pw[i] = 0.f; pb[i] = 0.f;
    const int e = min(i * 64 + lane, n - 1);      // slots past n re-read the last element and are masked afterwards
    const int c = e / F, f = e - c * F;
    off[i] = c * T * F + f;
    wv[i] = i * 64 + lane < n ? w[e] : 0.f;
  }
  const long rows = (long)B * T;
  for (long bt = (long)blockIdx.x * 4 + wave; bt < rows; bt += (long)gridDim.x * 4) {
    const int b = (int)(bt / T), t = (int)(bt - (long)b * T);
    const float mu = mean[bt], rs = rstd[bt];
    const long rowbase = (long)b * C * T * F + (long)t * F;
    // all loads of the row first, unconditionally (a guarded load per element is a branch and a wait per element)
    float g[LNCF_MAXV], xh[LNCF_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LNCF_MAXV; ++i) { g[i] = dy[rowbase + off[i]]; xh[i] = x[rowbase + off[i]]; }
#pragma unroll
    for (int i = 0; i < LNCF_MAXV; ++i) {
      const bool on = i * 64 + lane < n;
      const float d = on ? g[i] : 0.f;
      xh[i] = on ? (xh[i] - mu) * rs : 0.f;
      pw[i] += d * xh[i];
      pb[i] += d;
      g[i] = d * wv[i];
      s1 += g[i];
      s2 += g[i] * xh[i];
    }
    if (dx) {
      s1 = mpa_wave_sum(s1) / (float)n;
      s2 = mpa_wave_sum(s2) / (float)n;
#pragma unroll
      for (int i = 0; i < LNCF_MAXV; ++i)
        if (i * 64 + lane < n) dx[rowbase + off[i]] = rs * (g[i] - s1 - xh[i] * s2);
    }
  }
#pragma unroll
  for (int i = 0; i < LNCF_MAXV; ++i) {
    const int e = i * 64 + lane;
    if (e < n) { sh[(wave * 2 + 0) * n + e] = pw[i]; sh[(wave * 2 + 1) * n + e] = pb[i]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * n; e += 256) {
    const int which = e / n, k = e - which * n;
    float s = 0.f;
    for (int wv = 0; wv < 4; ++wv) s += sh[(wv * 2 + which) * n + k];
    ws[((long)blockIdx.x * 2 + which) * n + k] = s;
  }
}

// ws [S][2][n] -> o0[n], o1[n].  One workgroup per 64 outputs: lane = output, the four waves take every fourth partial
// row and are combined through LDS in a fixed order (deterministic, and S = 1024 rows are no longer one serial chain).
__global__ __launch_bounds__(256) void reduce2_kernel(const float* __restrict__ ws, float* __restrict__ o0,
                                                      float* __restrict__ o1, int n, int S) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  float s = 0.f;
  int which = 0, k = 0;
  if (i < 2 * n) {
    which = i / n; k = i - which * n;
    for (int j = wave; j < S; j += 4) s += ws[((long)j * 2 + which) * n + k];
  }
  part[wave][lane] = s;
  __syncthreads();
  if (wave == 0 && i < 2 * n) (which ? o1 : o0)[k] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}

// ---------------------------------------------------------------------------------- LayerNorm over rows (E <= 512)
constexpr int LNR_MAXV = 8;

__global__ __launch_bounds__(256) void layernorm_rows_fwd_kernel(const float* __restrict__ a, const float* __restrict__ r,
                                                                 const float* __restrict__ w, const float* __restrict__ bb,
                                                                 float* __restrict__ sum_out, float* __restrict__ y,
                                                                 float* __restrict__ mean, float* __restrict__ rstd,
                                                                 long rows, int E, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  float v[LNR_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LNR_MAXV; ++i) {
    const int e = i * 64 + lane;
    v[i] = 0.f;
    if (e < E) {
      v[i] = a[row * E + e] + (r ? r[row * E + e] : 0.f);
      if (sum_out) sum_out[row * E + e] = v[i];
      s += v[i];
    }
  }
  const float mu = mpa_wave_sum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LNR_MAXV; ++i) {
    const int e = i * 64 + lane;
    if (e < E) { const float d = v[i] - mu; q += d * d; }
  }
  const float rs = 1.0f / sqrtf(mpa_wave_sum(q) / (float)E + eps);
#pragma unroll
  for (int i = 0; i < LNR_MAXV; ++i) {
    const int e = i * 64 + lane;
    if (e < E) y[row * E + e] = (v[i] - mu) * rs * w[e] + bb[e];
  }
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

__global__ __launch_bounds__(256) void layernorm_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ xs,
                                                                 const float* __restrict__ w, const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, float* __restrict__ dx,
                                                                 float* __restrict__ ws, long rows, int E) {
  extern __shared__ float sh[];   // [4][2][E]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float pw[LNR_MAXV], pb[LNR_MAXV];
#pragma unroll
  for (int i = 0; i < LNR_MAXV; ++i) { pw[i] = 0.f; pb[i] = 0.f; }
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float mu = mean[row], rs = rstd[row];
    float g[LNR_MAXV], xh[LNR_MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < LNR_MAXV; ++i) {
      const int e = i * 64 + lane;
      g[i] = 0.f; xh[i] = 0.f;
      if (e < E) {
        const float d = dy[row * E + e];
        xh[i] = (xs[row * E + e] - mu) * rs;
        pw[i] += d * xh[i];
        pb[i] += d;
        g[i] = d * w[e];
        s1 += g[i];
        s2 += g[i] * xh[i];
      }
    }
    s1 = mpa_wave_sum(s1) / (float)E;
    s2 = mpa_wave_sum(s2) / (float)E;
#pragma unroll
    for (int i = 0; i < LNR_MAXV; ++i) {
      const int e = i * 64 + lane;
      if (e < E) dx[row * E + e] = rs * (g[i] - s1 - xh[i] * s2);
    }
  }
#pragma unroll
  for (int i = 0; i < LNR_MAXV; ++i) {
    const int e = i * 64 + lane;
    if (e < E) { sh[(wave * 2 + 0) * E + e] = pw[i]; sh[(wave * 2 + 1) * E + e] = pb[i]; }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 2 * E; e += 256) {
    const int which = e / E, k = e - which * E;
    float s = 0.f;
    for (int wv = 0; wv < 4; ++wv) s += sh[(wv * 2 + which) * E + k];
    ws[((long)blockIdx.x * 2 + which) * E + k] = s;
  }
}

// ---------------------------------------------------------------------------------- BatchNorm2d
// per-channel sums in double via atomics; grid (splits, C)
// The (b, r) position inside channel c advances incrementally (no per-element division) in units of V floats:
// V = 4 (16-byte loads) when HW % 4 == 0, else 1.
template <int V>
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, double* __restrict__ stats, int B, int C,
                                                       int HW) {
  const int c = blockIdx.y;
  const int HWv = HW / V;
  const unsigned per = (unsigned)B * HWv;
  const unsigned stride = gridDim.x * 256u;
  const unsigned sq = stride / HWv, sr = stride % HWv;
  unsigned i = blockIdx.x * 256u + threadIdx.x;
  unsigned b = i / HWv, r = i % HWv;
  float s = 0.f, q = 0.f;
  double ds = 0.0, dq = 0.0;
  int cnt = 0;
  for (; i < per; i += stride) {
    const float* ptr = x + ((long)b * C + c) * HW + (long)r * V;
    if (V == 4) {
      const float4 v = *reinterpret_cast<const float4*>(ptr);
      s += (v.x + v.y) + (v.z + v.w);
      q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    } else {
      const float v = *ptr;
      s += v;
      q += v * v;
    }
    if (++cnt == 16) { ds += s; dq += q; s = 0.f; q = 0.f; cnt = 0; }
    r += sr; b += sq;
    if (r >= (unsigned)HWv) { r -= HWv; b += 1; }
  }
  ds += s; dq += q;
  ds = mpa_wave_sum_d(ds);
  dq = mpa_wave_sum_d(dq);
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[wave * 2] = ds; sh[wave * 2 + 1] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[2 * c], sh[0] + sh[2] + sh[4] + sh[6]);
    atomicAdd(&stats[2 * c + 1], sh[1] + sh[3] + sh[5] + sh[7]);
  }
}

// grid (B*C, chunks): y = relu(gamma*(x-mu)*invstd + beta)
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const double* __restrict__ stats,
                                                       const float* __restrict__ rmean_in, const float* __restrict__ rvar_in,
                                                       float* __restrict__ y, int C, int HW, double count, float eps,
                                                       int relu) {
  const int plane = blockIdx.x, c = plane % C;
  float mu, invstd;
  if (stats) {
    const double m = stats[2 * c] / count;
    double var = stats[2 * c + 1] / count - m * m;
    if (var < 0) var = 0;
    mu = (float)m;
    invstd = (float)(1.0 / sqrt(var + (double)eps));
  } else {
    mu = rmean_in[c];
    invstd = 1.0f / sqrtf(rvar_in[c] + eps);
  }
  const float sc = gamma[c] * invstd, sf = beta[c] - mu * sc;
  const float* xp = x + (long)plane * HW;
  float* yp = y + (long)plane * HW;
  if ((HW & 3) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(xp);
    float4* y4 = reinterpret_cast<float4*>(yp);
    for (int i = blockIdx.y * 256 + threadIdx.x; i < HW / 4; i += gridDim.y * 256) {
      float4 v = x4[i];
      v.x = v.x * sc + sf; v.y = v.y * sc + sf; v.z = v.z * sc + sf; v.w = v.w * sc + sf;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      y4[i] = v;
    }
  } else {
    for (int i = blockIdx.y * 256 + threadIdx.x; i < HW; i += gridDim.y * 256) {
      float v = xp[i] * sc + sf;
      yp[i] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, float* running_mean, float* running_var,
                                   int64_t* nbt, float* save_mean, float* save_invstd, int C, double count, float momentum,
                                   float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= C) return;
  const double m = stats[2 * c] / count;
  double var = stats[2 * c + 1] / count - m * m;
  if (var < 0) var = 0;
  save_mean[c] = (float)m;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  const double unbiased = count > 1 ? var * count / (count - 1) : var;
  running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
  running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
}

// Batch statistics from the per-(pixel tile, channel) partial sums the producing convolution left behind
// (conv_fwd_kernel's epilogue): partials [rows][C][2] (sum, sum of squares; float) -> the same outputs as
// bn_finalize_kernel.  Two stages, both in a fixed order (run-to-run reproducible): bn_partials_stage1_kernel folds
// the rows into RS <= 64 double-precision rows (grid (C/16, RS): a batch-256 launch of the 16-channel layers has 5632
// rows, one workgroup per 16 channels took 43 us), bn_finalize_partials_kernel folds those and finalizes.
__global__ __launch_bounds__(256) void bn_partials_stage1_kernel(const float* __restrict__ partials, int rows, int C,
                                                                 double* __restrict__ out) {
  __shared__ double sh[16][16][2];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int RS = gridDim.y;
  const int per = (rows + RS - 1) / RS;
  const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
  double s = 0.0, q = 0.0;
  if (c < C) {
    for (int r = r0 + rg; r < r1; r += 16) {
      const float2 v = *reinterpret_cast<const float2*>(partials + ((long)r * C + c) * 2);
      s += (double)v.x;
      q += (double)v.y;
    }
  }
  sh[rg][cl][0] = s; sh[rg][cl][1] = q;
  __syncthreads();
  if (rg != 0 || c >= C) return;
  s = 0.0; q = 0.0;
  for (int j = 0; j < 16; ++j) { s += sh[j][cl][0]; q += sh[j][cl][1]; }
  out[((long)blockIdx.y * C + c) * 2] = s;
  out[((long)blockIdx.y * C + c) * 2 + 1] = q;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_finalize_partials_kernel(const T* __restrict__ partials, int rows,
                                                                   float* running_mean, float* running_var, int64_t* nbt,
                                                                   float* save_mean, float* save_invstd, int C, double count,
                                                                   float momentum, float eps) {
  __shared__ double sh[16][16][2];
  const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double s = 0.0, q = 0.0;
  if (c < C) {
    for (int r = rg; r < rows; r += 16) {
      s += (double)partials[((long)r * C + c) * 2];
      q += (double)partials[((long)r * C + c) * 2 + 1];
    }
  }
  sh[rg][cl][0] = s; sh[rg][cl][1] = q;
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
  if (rg != 0 || c >= C) return;
  s = 0.0; q = 0.0;
  for (int j = 0; j < 16; ++j) { s += sh[j][cl][0]; q += sh[j][cl][1]; }
  const double m = s / count;
  double var = q / count - m * m;
  if (var < 0) var = 0;
  save_mean[c] = (float)m;
  save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  const double unbiased = count > 1 ? var * count / (count - 1) : var;
  running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
  running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
}

// y = relu(gamma (x - mean) invstd + beta) from saved statistics; grid (B*C, chunks)
__global__ __launch_bounds__(256) void bn_apply_saved_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, float* __restrict__ y, int C,
                                                             int HW, int relu) {
  const int plane = blockIdx.x, c = plane % C;
  const float sc = gamma[c] * invstd[c], sf = beta[c] - mean[c] * sc;
  const float* xp = x + (long)plane * HW;
  float* yp = y + (long)plane * HW;
  if ((HW & 3) == 0) {
    const float4* x4 = reinterpret_cast<const float4*>(xp);
    float4* y4 = reinterpret_cast<float4*>(yp);
    for (int i = blockIdx.y * 256 + threadIdx.x; i < HW / 4; i += gridDim.y * 256) {
      float4 v = x4[i];
      v.x = v.x * sc + sf; v.y = v.y * sc + sf; v.z = v.z * sc + sf; v.w = v.w * sc + sf;
      if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      y4[i] = v;
    }
  } else {
    for (int i = blockIdx.y * 256 + threadIdx.x; i < HW; i += gridDim.y * 256) {
      float v = xp[i] * sc + sf;
      yp[i] = relu ? fmaxf(v, 0.f) : v;
    }
  }
}

__global__ void bn_eval_save_kernel(const float* rmean, const float* rvar, float* save_mean, float* save_invstd, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  save_mean[c] = rmean[c];
  save_invstd[c] = 1.0f / sqrtf(rvar[c] + eps);
}

// backward sums: S1 = sum g, S2 = sum g*xhat, g = dy * (y>0)
// ReLU mask: with `beta` given it is recomputed from x exactly as bn_apply_kernel formed y (x*sc + sf > 0, same float
// operations), so y is never read; without beta it is y > 0.
// PART: the block leaves its pair of sums in stats[(c * gridDim.x + blockIdx.x) * 2 ..] (bn_bwd_apply_kernel adds the
// gridDim.x <= 64 pairs of its channel in a fixed order: no zero-fill launch, no atomics, run-to-run reproducible);
// otherwise the blocks add into the zeroed stats[2 c ..] atomically (the SyncBN halves, which hand 2 C sums to the caller).
template <int V, bool PART>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd, double* __restrict__ stats,
                                                           int B, int C, int HW, int relu) {
  const int c = blockIdx.y;
  const int HWv = HW / V;
  const unsigned per = (unsigned)B * HWv;
  const unsigned stride = gridDim.x * 256u;
  const unsigned sq = stride / HWv, sr = stride % HWv;
  unsigned i = blockIdx.x * 256u + threadIdx.x;
  unsigned b = i / HWv, r = i % HWv;
  const float mu = save_mean[c], is = save_invstd[c];
  const bool from_x = beta != nullptr;
  const float sc = gamma[c] * is, sf = from_x ? beta[c] - mu * sc : 0.f;
  float s = 0.f, q = 0.f;
  double ds = 0.0, dq = 0.0;
  int cnt = 0;
  for (; i < per; i += stride) {
    const long o = ((long)b * C + c) * HW + (long)r * V;
    if (V == 4) {
      float4 g = *reinterpret_cast<const float4*>(dy + o);
      const float4 xv = *reinterpret_cast<const float4*>(x + o);
      if (relu) {
        float4 yv;
        if (from_x) { yv.x = xv.x * sc + sf; yv.y = xv.y * sc + sf; yv.z = xv.z * sc + sf; yv.w = xv.w * sc + sf; }
        else yv = *reinterpret_cast<const float4*>(y + o);
        if (!(yv.x > 0.f)) g.x = 0.f;
        if (!(yv.y > 0.f)) g.y = 0.f;
        if (!(yv.z > 0.f)) g.z = 0.f;
        if (!(yv.w > 0.f)) g.w = 0.f;
      }
      s += (g.x + g.y) + (g.z + g.w);
      q += (g.x * (xv.x - mu) * is + g.y * (xv.y - mu) * is) + (g.z * (xv.z - mu) * is + g.w * (xv.w - mu) * is);
    } else {
      float g = dy[o];
      if (relu && !((from_x ? x[o] * sc + sf : y[o]) > 0.f)) g = 0.f;
      s += g;
      q += g * (x[o] - mu) * is;
    }
    if (++cnt == 16) { ds += s; dq += q; s = 0.f; q = 0.f; cnt = 0; }
    r += sr; b += sq;
    if (r >= (unsigned)HWv) { r -= HWv; b += 1; }
  }
  ds += s; dq += q;
  ds = mpa_wave_sum_d(ds);
  dq = mpa_wave_sum_d(dq);
  __shared__ double sh[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { sh[wave * 2] = ds; sh[wave * 2 + 1] = dq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (PART) {
      double* o = stats + ((long)c * gridDim.x + blockIdx.x) * 2;
      o[0] = sh[0] + sh[2] + sh[4] + sh[6];
      o[1] = sh[1] + sh[3] + sh[5] + sh[7];
    } else {
      atomicAdd(&stats[2 * c], sh[0] + sh[2] + sh[4] + sh[6]);
      atomicAdd(&stats[2 * c + 1], sh[1] + sh[3] + sh[5] + sh[7]);
    }
  }
}

// nsplit > 0: stats holds nsplit <= 64 partial pairs per channel (bn_bwd_stats_kernel<V, true>); every wave adds its
// channel's pairs itself (one pair per lane, butterfly sum: the same value in every wave of every block)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           const float* __restrict__ y, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ save_mean,
                                                           const float* __restrict__ save_invstd,
                                                           const double* __restrict__ stats, float* __restrict__ dx, int C,
                                                           int HW, double count, int relu, int train,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int nsplit) {
  const int plane = blockIdx.x, c = plane % C;
  double s1, s2;
  if (nsplit > 0) {
    const int lane = threadIdx.x & 63;
    const double* pp = stats + ((long)c * nsplit + (lane < nsplit ? lane : 0)) * 2;
    s1 = mpa_wave_sum_d(lane < nsplit ? pp[0] : 0.0);
    s2 = mpa_wave_sum_d(lane < nsplit ? pp[1] : 0.0);
  } else {
    s1 = stats[2 * c];
    s2 = stats[2 * c + 1];
  }
  // the parameter gradients are the two sums themselves: written here by the first image's blocks instead of by an 18th-of-a-
  // step launch of their own (dgamma != nullptr)
  if (dgamma != nullptr && plane < C && blockIdx.y == 0 && threadIdx.x == 0) {
    dbeta[c] = (float)s1;
    dgamma[c] = (float)s2;
  }
  const float mu = save_mean[c], is = save_invstd[c];
  const float k = gamma[c] * is;
  const bool from_x = beta != nullptr;              // ReLU mask recomputed from x (see bn_bwd_stats_kernel)
  const float sf = from_x ? beta[c] - mu * k : 0.f;
  const float m1 = train ? (float)(s1 / count) : 0.f;
  const float m2 = train ? (float)(s2 / count) : 0.f;
  const long base = (long)plane * HW;
  if ((HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dx) |
                         reinterpret_cast<uintptr_t>(y)) & 15) == 0) {
    const float4* g4 = reinterpret_cast<const float4*>(dy + base);
    const float4* x4 = reinterpret_cast<const float4*>(x + base);
    const float4* y4 = reinterpret_cast<const float4*>(y + base);
    float4* d4 = reinterpret_cast<float4*>(dx + base);
    for (int i = blockIdx.y * 256 + threadIdx.x; i < HW / 4; i += gridDim.y * 256) {
      float4 g = g4[i];
      const float4 xv = x4[i];
      if (relu) {
        float4 yv;
        if (from_x) { yv.x = xv.x * k + sf; yv.y = xv.y * k + sf; yv.z = xv.z * k + sf; yv.w = xv.w * k + sf; }
        else yv = y4[i];
        if (!(yv.x > 0.f)) g.x = 0.f;
        if (!(yv.y > 0.f)) g.y = 0.f;
        if (!(yv.z > 0.f)) g.z = 0.f;
        if (!(yv.w > 0.f)) g.w = 0.f;
      }
      float4 o;
      o.x = k * (g.x - m1 - (xv.x - mu) * is * m2);
      o.y = k * (g.y - m1 - (xv.y - mu) * is * m2);
      o.z = k * (g.z - m1 - (xv.z - mu) * is * m2);
      o.w = k * (g.w - m1 - (xv.w - mu) * is * m2);
      d4[i] = o;
    }
    return;
  }
  for (int i = blockIdx.y * 256 + threadIdx.x; i < HW; i += gridDim.y * 256) {
    float g = dy[base + i];
    const float xv = x[base + i];
    if (relu && !((from_x ? xv * k + sf : y[base + i]) > 0.f)) g = 0.f;
    const float xh = (xv - mu) * is;
    dx[base + i] = k * (g - m1 - xh * m2);
  }
}

// the partial pairs of bn_bwd_stats_kernel<V, true> summed once per channel (one wave per channel, same butterfly order as
// the in-kernel sum of bn_bwd_apply_kernel): for tensors whose apply launch has so many workgroups that each of them
// re-adding the 64 pairs costs more than this 5-us launch
__global__ __launch_bounds__(64) void bn_bwd_sum_partials_kernel(const double* __restrict__ part, int nsplit,
                                                                 double* __restrict__ stats) {
  const int c = blockIdx.x, lane = threadIdx.x;
  const double* pp = part + ((long)c * nsplit + (lane < nsplit ? lane : 0)) * 2;
  const double s1 = mpa_wave_sum_d(lane < nsplit ? pp[0] : 0.0);
  const double s2 = mpa_wave_sum_d(lane < nsplit ? pp[1] : 0.0);
  if (lane == 0) { stats[2 * c] = s1; stats[2 * c + 1] = s2; }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ stats, float* dgamma, float* dbeta, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  dbeta[c] = (float)stats[2 * c];
  dgamma[c] = (float)stats[2 * c + 1];
}

#define MPA_BN_BWD_SPLITS 64      // partial pairs per channel bn_bwd_apply_kernel sums (one per lane)
#define MPA_BN_BWD_INLINE_BLOCKS 8192   // apply launches with more workgroups than this get the pairs summed by a launch of its own
inline int stat_splits(int B, int C, int HW) {
  long per = (long)B * HW;
  long want = std::max<long>(1, (256L * 16) / C);
  long maxs = std::max<long>(1, per / 4096);
  return (int)std::max<long>(1, std::min(want, maxs));
}

}  // namespace

extern "C" {

int mpa_layernorm_cf_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd, int B, int C,
                         int T, int F, float eps, void* stream) {
  if (!x || !w || !b || !y || !mean || !rstd || C * F > 64 * LNCF_MAXV) return MPA_ERR_ARG;
  const long rows = (long)B * T;
  MPA_LAUNCH(layernorm_cf_fwd_kernel, dim3((unsigned)mpa_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, w, b,
                     y, mean, rstd, B, C, T, F, eps);
  return mpa_launch_status();
}

// workspace (caller-allocated, mpa_layernorm_bwd_workspace bytes): one partial dgamma/dbeta row per workgroup, reduced
// by reduce2_kernel in a fixed order
#define LN_BWD_BLOCKS 1024   // at most this many workgroups in the row loop (each leaves one partial dgamma/dbeta row)
int64_t mpa_layernorm_bwd_workspace(int n) { return (int64_t)LN_BWD_BLOCKS * 2 * n * 4; }
// eight rows per wave at least: with few rows (local batch 32: 2400 rows) 1024 partial rows made the fixed-order
// reduction of the partials cost more than the row loop itself
static inline int ln_bwd_blocks(int64_t rows) {
  const int64_t b = mpa_cdiv(rows, 32);
  return (int)(b < 32 ? 32 : b > LN_BWD_BLOCKS ? LN_BWD_BLOCKS : b);
}

int mpa_layernorm_cf_bwd_ws(const float* dy, const float* x, const float* w, const float* mean, const float* rstd, float* dx,
                            float* dw, float* db, void* ws, int B, int C, int T, int F, void* stream) {
  if (!dy || !x || !w || !mean || !rstd || !dw || !db || !ws || C * F > 64 * LNCF_MAXV) return MPA_ERR_ARG;
  const int n = C * F;
  hipStream_t s = (hipStream_t)stream;
  const int nb = ln_bwd_blocks((int64_t)B * T);
  MPA_LAUNCH(layernorm_cf_bwd_kernel, dim3(nb), dim3(256), (size_t)8 * n * 4, s, dy, x, w, mean, rstd, dx,
                     (float*)ws, B, C, T, F);
  int rc = mpa_launch_status();
  if (rc) return rc;
  MPA_LAUNCH(reduce2_kernel, dim3((unsigned)mpa_cdiv(2 * n, 64)), dim3(256), 0, s, (const float*)ws, dw, db, n, nb);
  return mpa_launch_status();
}

int mpa_layernorm_rows_fwd(const float* a, const float* r, const float* w, const float* b, float* sum_out, float* y,
                           float* mean, float* rstd, int64_t rows, int E, float eps, void* stream) {
  if (!a || !w || !b || !y || !mean || !rstd || E > 64 * LNR_MAXV) return MPA_ERR_ARG;
  MPA_LAUNCH(layernorm_rows_fwd_kernel, dim3((unsigned)mpa_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, a, r, w,
                     b, sum_out, y, mean, rstd, (long)rows, E, eps);
  return mpa_launch_status();
}

int mpa_layernorm_rows_bwd_ws(const float* dy, const float* xs, const float* w, const float* mean, const float* rstd,
                              float* dx, float* dw, float* db, void* ws, int64_t rows, int E, void* stream) {
  if (!dy || !xs || !w || !mean || !rstd || !dx || !dw || !db || !ws || E > 64 * LNR_MAXV) return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int nb = ln_bwd_blocks(rows);
  MPA_LAUNCH(layernorm_rows_bwd_kernel, dim3(nb), dim3(256), (size_t)8 * E * 4, s, dy, xs, w, mean, rstd,
                     dx, (float*)ws, (long)rows, E);
  int rc = mpa_launch_status();
  if (rc) return rc;
  MPA_LAUNCH(reduce2_kernel, dim3((unsigned)mpa_cdiv(2 * E, 64)), dim3(256), 0, s, (const float*)ws, dw, db, E, nb);
  return mpa_launch_status();
}

int mpa_bn_relu_train_fwd(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          int64_t* num_batches_tracked, float* y, float* save_mean, float* save_invstd, double* stats_ws,
                          int B, int C, int HW, float momentum, float eps, int relu, void* stream) {
  if (!x || !gamma || !beta || !running_mean || !running_var || !y || !save_mean || !save_invstd || !stats_ws)
    return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (mpa_zero_async(stats_ws, sizeof(double) * 2 * C, s) != MPA_OK) return MPA_ERR_LAUNCH;
  if ((long)B * HW > 0x7fffffffL) return MPA_ERR_ARG;
  const int splits = stat_splits(B, C, HW);
  const bool vec = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (vec) MPA_LAUNCH(bn_stats_kernel<4>, dim3(splits, C), dim3(256), 0, s, x, stats_ws, B, C, HW);
  else MPA_LAUNCH(bn_stats_kernel<1>, dim3(splits, C), dim3(256), 0, s, x, stats_ws, B, C, HW);
  const double count = (double)B * HW;
  const int chunks = (int)std::max<long>(1, std::min<long>(mpa_cdiv(HW, 1024), 64));
  MPA_LAUNCH(bn_apply_kernel, dim3(B * C, chunks), dim3(256), 0, s, x, gamma, beta, (const double*)stats_ws,
                     (const float*)nullptr, (const float*)nullptr, y, C, HW, count, eps, relu);
  MPA_LAUNCH(bn_finalize_kernel, dim3((unsigned)mpa_cdiv(C, 64)), dim3(64), 0, s, (const double*)stats_ws,
                     running_mean, running_var, num_batches_tracked, save_mean, save_invstd, C, count, momentum, eps);
  return mpa_launch_status();
}

int mpa_bn_relu_train_fwd_partials(const float* x, const float* partials, int rows, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                                   float* save_mean, float* save_invstd, double* stage_ws, int B, int C, int HW,
                                   float momentum, float eps, int relu, void* stream) {
  if (!x || !partials || rows <= 0 || !gamma || !beta || !running_mean || !running_var || !y || !save_mean || !save_invstd)
    return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (rows > 512 && stage_ws) {           // stage_ws: 64 * C * 2 doubles
    const int RS = (int)std::min<long>(64, mpa_cdiv(rows, 64));
    MPA_LAUNCH(bn_partials_stage1_kernel, dim3((unsigned)mpa_cdiv(C, 16), (unsigned)RS), dim3(256), 0, s, partials, rows, C,
               stage_ws);
    if ((rc = mpa_launch_status()) != MPA_OK) return rc;
    MPA_LAUNCH(bn_finalize_partials_kernel<double>, dim3((unsigned)mpa_cdiv(C, 16)), dim3(256), 0, s,
               (const double*)stage_ws, RS, running_mean, running_var, num_batches_tracked, save_mean, save_invstd, C,
               (double)B * HW, momentum, eps);
  } else {
    MPA_LAUNCH(bn_finalize_partials_kernel<float>, dim3((unsigned)mpa_cdiv(C, 16)), dim3(256), 0, s, partials, rows,
               running_mean, running_var, num_batches_tracked, save_mean, save_invstd, C, (double)B * HW, momentum, eps);
  }
  if ((rc = mpa_launch_status()) != MPA_OK) return rc;
  const int chunks = (int)std::max<long>(1, std::min<long>(mpa_cdiv(HW, 1024), 64));
  MPA_LAUNCH(bn_apply_saved_kernel, dim3(B * C, chunks), dim3(256), 0, s, x, gamma, beta, (const float*)save_mean,
             (const float*)save_invstd, y, C, HW, relu);
  return mpa_launch_status();
}

int mpa_bn_relu_eval_fwd(const float* x, const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float* y, float* save_mean, float* save_invstd, int B, int C, int HW,
                         float eps, int relu, void* stream) {
  if (!x || !gamma || !beta || !running_mean || !running_var || !y) return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = (int)std::max<long>(1, std::min<long>(mpa_cdiv(HW, 1024), 64));
  MPA_LAUNCH(bn_apply_kernel, dim3(B * C, chunks), dim3(256), 0, s, x, gamma, beta, (const double*)nullptr,
                     running_mean, running_var, y, C, HW, 1.0, eps, relu);
  if (save_mean && save_invstd)
    MPA_LAUNCH(bn_eval_save_kernel, dim3((unsigned)mpa_cdiv(C, 64)), dim3(64), 0, s, running_mean, running_var,
                       save_mean, save_invstd, C, eps);
  return mpa_launch_status();
}

int mpa_bn_relu_bwd(const float* dy, const float* x, const float* y, const float* gamma, const float* beta,
                    const float* save_mean, const float* save_invstd, float* dx, float* dgamma, float* dbeta,
                    double* stats_ws, int B, int C, int HW, int relu, int train, void* stream) {
  if (!dy || !x || !gamma || !save_mean || !save_invstd || !dx || !dgamma || !dbeta || !stats_ws || (relu && !y && !beta))
    return MPA_ERR_ARG;
  if (!y) y = x;      // never dereferenced when beta is given; keeps the alignment test below meaningful
  hipStream_t s = (hipStream_t)stream;
  if ((long)B * HW > 0x7fffffffL) return MPA_ERR_ARG;
  // two launches, no zero-fill: at most MPA_BN_BWD_SPLITS partial pairs per channel, summed by the apply kernel's waves
  const int chunks = (int)std::max<long>(1, std::min<long>(mpa_cdiv(HW, 1024), 64));
  // many apply workgroups: the pairs are summed once by a launch of its own (into the last 2 C doubles of the workspace,
  // which the partials of at most 63 splits leave free) instead of once per apply workgroup
  const bool big = (long)B * C * chunks > MPA_BN_BWD_INLINE_BLOCKS;
  const int splits = std::min(stat_splits(B, C, HW), big ? MPA_BN_BWD_SPLITS - 1 : MPA_BN_BWD_SPLITS);
  const bool vec = HW % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy) |
                                    reinterpret_cast<uintptr_t>(y)) & 15) == 0;
  if (vec) MPA_LAUNCH((bn_bwd_stats_kernel<4, true>), dim3(splits, C), dim3(256), 0, s, dy, x, y, gamma, beta, save_mean,
                      save_invstd, stats_ws, B, C, HW, relu);
  else MPA_LAUNCH((bn_bwd_stats_kernel<1, true>), dim3(splits, C), dim3(256), 0, s, dy, x, y, gamma, beta, save_mean, save_invstd,
                  stats_ws, B, C, HW, relu);
  if (big) {
    double* sums = stats_ws + (long)2 * C * (MPA_BN_BWD_SPLITS - 1);
    MPA_LAUNCH(bn_bwd_sum_partials_kernel, dim3(C), dim3(64), 0, s, (const double*)stats_ws, splits, sums);
    MPA_LAUNCH(bn_bwd_apply_kernel, dim3(B * C, chunks), dim3(256), 0, s, dy, x, y, gamma, beta, save_mean, save_invstd,
               (const double*)sums, dx, C, HW, (double)B * HW, relu, train, dgamma, dbeta, 0);
    return mpa_launch_status();
  }
  MPA_LAUNCH(bn_bwd_apply_kernel, dim3(B * C, chunks), dim3(256), 0, s, dy, x, y, gamma, beta, save_mean, save_invstd,
                     (const double*)stats_ws, dx, C, HW, (double)B * HW, relu, train, dgamma, dbeta, splits);
  return mpa_launch_status();
}

// ---- the same passes in two halves, with the per-channel sums handed to the caller in between: a data-parallel rank
// all-reduces them over the ranks (SyncBN, SURVEY 8e's optional exactness mode) before the second half
int mpa_bn_batch_sums(const float* x, double* sums, int B, int C, int HW, void* stream) {
  if (!x || !sums || (long)B * HW > 0x7fffffffL) return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (mpa_zero_async(sums, sizeof(double) * 2 * C, s) != MPA_OK) return MPA_ERR_LAUNCH;
  const int splits = stat_splits(B, C, HW);
  const bool vec = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
  if (vec) MPA_LAUNCH(bn_stats_kernel<4>, dim3(splits, C), dim3(256), 0, s, x, sums, B, C, HW);
  else MPA_LAUNCH(bn_stats_kernel<1>, dim3(splits, C), dim3(256), 0, s, x, sums, B, C, HW);
  return mpa_launch_status();
}

int mpa_bn_relu_train_fwd_sums(const float* x, const double* sums, double count, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, int64_t* num_batches_tracked, float* y,
                               float* save_mean, float* save_invstd, int B, int C, int HW, float momentum, float eps, int relu,
                               void* stream) {
  if (!x || !sums || !gamma || !beta || !running_mean || !running_var || !y || !save_mean || !save_invstd || count <= 0)
    return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = (int)std::max<long>(1, std::min<long>(mpa_cdiv(HW, 1024), 64));
  MPA_LAUNCH(bn_apply_kernel, dim3(B * C, chunks), dim3(256), 0, s, x, gamma, beta, sums, (const float*)nullptr,
             (const float*)nullptr, y, C, HW, count, eps, relu);
  MPA_LAUNCH(bn_finalize_kernel, dim3((unsigned)mpa_cdiv(C, 64)), dim3(64), 0, s, sums, running_mean, running_var,
             num_batches_tracked, save_mean, save_invstd, C, count, momentum, eps);
  return mpa_launch_status();
}

int mpa_bn_relu_bwd_sums(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, double* stats, float* dgamma, float* dbeta, int B, int C, int HW, int relu,
                         void* stream) {
  if (!dy || !x || !gamma || !beta || !save_mean || !save_invstd || !stats || !dgamma || !dbeta || (long)B * HW > 0x7fffffffL)
    return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (mpa_zero_async(stats, sizeof(double) * 2 * C, s) != MPA_OK) return MPA_ERR_LAUNCH;
  const int splits = stat_splits(B, C, HW);
  const bool vec = HW % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15) == 0;
  if (vec) MPA_LAUNCH((bn_bwd_stats_kernel<4, false>), dim3(splits, C), dim3(256), 0, s, dy, x, x, gamma, beta, save_mean, save_invstd,
                      stats, B, C, HW, relu);
  else MPA_LAUNCH((bn_bwd_stats_kernel<1, false>), dim3(splits, C), dim3(256), 0, s, dy, x, x, gamma, beta, save_mean, save_invstd, stats,
                  B, C, HW, relu);
  // this rank's parameter gradients come from its own sums (the gradient averager sums them over the ranks)
  MPA_LAUNCH(bn_bwd_finalize_kernel, dim3((unsigned)mpa_cdiv(C, 64)), dim3(64), 0, s, (const double*)stats, dgamma, dbeta, C);
  return mpa_launch_status();
}

int mpa_bn_relu_bwd_apply(const float* dy, const float* x, const float* gamma, const float* beta, const float* save_mean,
                          const float* save_invstd, const double* stats, double count, float* dx, int B, int C, int HW,
                          int relu, void* stream) {
  if (!dy || !x || !gamma || !beta || !save_mean || !save_invstd || !stats || !dx || count <= 0) return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int chunks = (int)std::max<long>(1, std::min<long>(mpa_cdiv(HW, 1024), 64));
  MPA_LAUNCH(bn_bwd_apply_kernel, dim3(B * C, chunks), dim3(256), 0, s, dy, x, x, gamma, beta, save_mean, save_invstd, stats,
             dx, C, HW, count, relu, 1, (float*)nullptr, (float*)nullptr, 0);
  return mpa_launch_status();
}

}  // extern "C"
