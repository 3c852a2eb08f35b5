// Pointwise / small-reduction kernels: activations, dropout, adds, transposes, channel sums, losses,
// LSTM cell, AdamW.  All HBM-bound; float4 where alignment allows.
#include "mpa_common.h"
#include <algorithm>

__global__ __launch_bounds__(256) void mpa_zero_kernel(uint32_t* __restrict__ p, long rows, long row_words, long pitch_words) {
  const long n = rows * row_words;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / row_words, c = i - r * row_words;
    p[r * pitch_words + c] = 0u;
  }
}
int mpa_zero2d_async(void* ptr, size_t pitch_bytes, size_t width_bytes, size_t rows, hipStream_t s) {
  if ((pitch_bytes | width_bytes | (size_t)(uintptr_t)ptr) & 3) return MPA_ERR_ARG;
  if (!width_bytes || !rows) return MPA_OK;
  const long n = (long)(rows * (width_bytes / 4));
  const long blocks = n / 256 + 1 < 2048 ? n / 256 + 1 : 2048;
  MPA_LAUNCH(mpa_zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (uint32_t*)ptr, (long)rows, (long)(width_bytes / 4),
             (long)(pitch_bytes / 4));
  return mpa_launch_status();
}
int mpa_zero_async(void* ptr, size_t bytes, hipStream_t s) { return mpa_zero2d_async(ptr, bytes, bytes, 1, s); }

namespace {

inline unsigned blocks_for(long n, int per = 256) {
  return (unsigned)std::max<long>(1, std::min<long>(mpa_cdiv(n, per), 1 << 16));
}

__global__ __launch_bounds__(256) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int act,
                                                      float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    float r;
    if (act == MPA_ACT_SIGMOID) r = 1.f / (1.f + expf(-v));
    else if (act == MPA_ACT_SELU) r = 1.0507009873554805f * (v > 0.f ? v : 1.6732632423543772f * expm1f(v));
    else r = mpa_apply_act(v, act, slope);
    y[i] = r;
  }
}

// dx = dy * act'(x); for sigmoid `x` must be the *output* y
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                      float* __restrict__ dx, long n, int act, float slope) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i], g = dy[i];
    float r;
    switch (act) {
      case MPA_ACT_RELU: r = v > 0.f ? g : 0.f; break;
      case MPA_ACT_LRELU: r = v >= 0.f ? g : g * slope; break;
      case MPA_ACT_SIGMOID: r = g * v * (1.f - v); break;
      case MPA_ACT_ELU: r = v > 0.f ? g : g * (v + 1.f); break;      // v = y: y <= 0 <=> x <= 0, dy/dx = exp(x) = y + 1
      case MPA_ACT_SELU: r = v > 0.f ? g * 1.0507009873554805f : g * (v + 1.0507009873554805f * 1.6732632423543772f); break;   // v = y
      default: r = g;
    }
    dx[i] = r;
  }
}

// counter-based RNG (splitmix64 finaliser over seed/offset/index); keep iff u >= p, scale 1/(1-p)
__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t idx) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}

// rng_state (device): [0] seed, [1] base offset of the current step; `offset` is the call's position inside the step.
// Both live on the device so that a captured HIP graph of the training step draws fresh masks at every replay
// (mpa_u64_add advances the base at the end of the step).
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float p,
                                                      float scale, const uint64_t* __restrict__ rng_state, uint64_t local) {
  const uint64_t seed = rng_state[0], offset = rng_state[1] + local;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float u = rng_uniform(seed, offset + (uint64_t)i);
    y[i] = u >= p ? x[i] * scale : 0.f;
  }
}

// MaxPool2d((KH,1), stride 1, padding (KH/2, 0)) -> Dropout [-> + residual] in one pass: the tail of the CNN families'
// prefilter stages (KH = 3, basic_cnns.py:374-377, with the residual add of deep_cnn_segm_sigmoid.forward, :414-418) and
// of every model's head stage conv2 (KH = 13, basic_cnns.py:380-385 / unet_cnns.py:538-543).  Instead of three passes
// (pool with an int32 argmax plane, dropout, add: 2.6 GB of traffic per 290 MB tensor) the activations cross HBM once
// each way (0.94 GB).  A thread owns V adjacent columns and R consecutive rows: it loads the R + KH - 1 input rows once
// and forms the R windows in registers.  Same rules as the separate kernels, bit for bit: first maximum wins and a NaN
// propagates (maxpool_fwd_plane_kernel), the keep mask is rng_uniform(seed, base + offset + flat index) >= p
// (dropout_kernel), and the sum is (pooled * scale) + residual.  `which` records the window row (0..KH-1) of the maximum and whether it is positive.
constexpr int POOLROWS_NONPOS = 0x40;     // flag in `which`: the window's maximum is <= 0 (or NaN) -- for the fused activation backward

template <int KH, int R, int V>
__global__ __launch_bounds__(256) void poolrows_drop_add_fwd_kernel(const float* __restrict__ h, const float* __restrict__ res,
                                                                    float* __restrict__ out, int8_t* __restrict__ which,
                                                                    long planes, int H, int W, float p, float scale,
                                                                    const uint64_t* __restrict__ rng_state, uint64_t local) {
  constexpr int PAD = KH / 2;
  const int WV = W / V, nblk = (H + R - 1) / R;
  const long items = planes * nblk * WV;
  const bool drop = p > 0.f;
  const uint64_t seed = drop ? rng_state[0] : 0, offset = drop ? rng_state[1] + local : 0;
  for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < items; item += (long)gridDim.x * 256) {
    const long pb = item / WV;
    const int col = (int)(item - pb * WV) * V;
    const long pl = pb / nblk;
    const int r0 = (int)(pb - pl * nblk) * R;
    const long base = pl * H * W + col;
    // All loads first and unconditional: rows outside the plane read a clamped row and are never looked at.  (A guarded
    // load per row compiles to a branch and a wait per row, i.e. R + KH - 1 serial trips to memory per item.)
    float v[R + KH - 1][V], rv[R][V];
#pragma unroll
    for (int j = 0; j < R + KH - 1; ++j) {
      const int row = min(max(r0 - PAD + j, 0), H - 1);
      if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(h + base + (long)row * W);
        v[j][0] = t.x; v[j][1] = t.y; v[j][2] = t.z; v[j][3] = t.w;
      } else if constexpr (V == 2) {
        const float2 t = *reinterpret_cast<const float2*>(h + base + (long)row * W);
        v[j][0] = t.x; v[j][1] = t.y;
      } else {
        v[j][0] = h[base + (long)row * W];
      }
    }
    if (res) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = min(r0 + r, H - 1);
        if constexpr (V == 4) {
          const float4 t = *reinterpret_cast<const float4*>(res + base + (long)row * W);
          rv[r][0] = t.x; rv[r][1] = t.y; rv[r][2] = t.z; rv[r][3] = t.w;
        } else if constexpr (V == 2) {
          const float2 t = *reinterpret_cast<const float2*>(res + base + (long)row * W);
          rv[r][0] = t.x; rv[r][1] = t.y;
        } else {
          rv[r][0] = res[base + (long)row * W];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = r0 + r;
      if (row >= H) continue;
      const long e = base + (long)row * W;
      float o[V];
      int8_t wsel[V];
#pragma unroll
      for (int c = 0; c < V; ++c) {
        float best = 0.f;
        int sel = -1;
#pragma unroll
        for (int d = 0; d < KH; ++d) {
          const int iy = row - PAD + d;
          if (iy < 0 || iy >= H) continue;
          const float t = v[r + d][c];
          if (sel < 0 || t > best || t != t) { best = t; sel = d; }
        }
        float y = best;
        if (drop) y = rng_uniform(seed, offset + (uint64_t)(e + c)) >= p ? best * scale : 0.f;
        o[c] = res ? y + rv[r][c] : y;
        wsel[c] = (int8_t)(sel | (best > 0.f ? 0 : POOLROWS_NONPOS));     // bit 6: the maximum is not positive
      }
      if constexpr (V == 4) {
        *reinterpret_cast<float4*>(out + e) = float4{o[0], o[1], o[2], o[3]};
        if (which) *reinterpret_cast<char4*>(which + e) = char4{wsel[0], wsel[1], wsel[2], wsel[3]};
      } else if constexpr (V == 2) {
        *reinterpret_cast<float2*>(out + e) = float2{o[0], o[1]};
        if (which) *reinterpret_cast<char2*>(which + e) = char2{wsel[0], wsel[1]};
      } else {
        out[e] = o[0];
        if (which) which[e] = wsel[0];
      }
    }
  }
}

// Backward of the pool + dropout part (the residual's gradient is dout itself): gather form, fixed order -- input row
// rho collects g(r) = keep(r) * scale * dout(r) from the windows r = rho + PAD ... rho - PAD whose recorded row is rho.
// neg_slope != 1: h is the output of a ReLU / LeakyReLU that the producing convolution applied in its epilogue, and that
// activation's backward pass is folded in: an element that won a window *is* that window's maximum, so its sign is the window's
// flag, and the collected sum is multiplied by neg_slope where it is not positive (= mpa_act_bwd on the result, bit for bit).
template <int KH, int R, int V>
__global__ __launch_bounds__(256) void poolrows_drop_bwd_kernel(const float* __restrict__ dout, const int8_t* __restrict__ which,
                                                                float* __restrict__ dh, long planes, int H, int W, float p,
                                                                float scale, const uint64_t* __restrict__ rng_state,
                                                                uint64_t local, float neg_slope) {
  constexpr int PAD = KH / 2;
  const int WV = W / V, nblk = (H + R - 1) / R;
  const long items = planes * nblk * WV;
  const bool drop = p > 0.f;
  const uint64_t seed = drop ? rng_state[0] : 0, offset = drop ? rng_state[1] + local : 0;
  for (long item = (long)blockIdx.x * 256 + threadIdx.x; item < items; item += (long)gridDim.x * 256) {
    const long pb = item / WV;
    const int col = (int)(item - pb * WV) * V;
    const long pl = pb / nblk;
    const int r0 = (int)(pb - pl * nblk) * R;
    const long base = pl * H * W + col;
    float g[R + KH - 1][V];
    int8_t w[R + KH - 1][V];
#pragma unroll
    for (int j = 0; j < R + KH - 1; ++j) {      // all loads first, unconditional (clamped rows are masked below)
      const int row = min(max(r0 - PAD + j, 0), H - 1);
      const long e = base + (long)row * W;
      if constexpr (V == 4) {
        const float4 t = *reinterpret_cast<const float4*>(dout + e);
        const char4 s = *reinterpret_cast<const char4*>(which + e);
        g[j][0] = t.x; g[j][1] = t.y; g[j][2] = t.z; g[j][3] = t.w;
        w[j][0] = s.x; w[j][1] = s.y; w[j][2] = s.z; w[j][3] = s.w;
      } else if constexpr (V == 2) {
        const float2 t = *reinterpret_cast<const float2*>(dout + e);
        const char2 s = *reinterpret_cast<const char2*>(which + e);
        g[j][0] = t.x; g[j][1] = t.y;
        w[j][0] = s.x; w[j][1] = s.y;
      } else {
        g[j][0] = dout[e]; w[j][0] = which[e];
      }
    }
#pragma unroll
    for (int j = 0; j < R + KH - 1; ++j) {
      const int row = r0 - PAD + j;      // window (= output) row
      const bool on = row >= 0 && row < H;
      const long e = base + (long)row * W;
#pragma unroll
      for (int c = 0; c < V; ++c) {
        float t = on ? g[j][c] : 0.f;
        if (drop) t = rng_uniform(seed, offset + (uint64_t)(e + c)) >= p ? t * scale : 0.f;
        g[j][c] = t;
        w[j][c] = on ? w[j][c] : (int8_t)-1;
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int row = r0 + r;
      if (row >= H) continue;
      float o[V];
#pragma unroll
      for (int c = 0; c < V; ++c) {
        // window rho + PAD - d holds rho as its row d (register row r + 2 PAD - d)
        float a = 0.f;
        bool nonpos = false;
#pragma unroll
        for (int d = 0; d < KH; ++d) {
          const int wv = w[r + 2 * PAD - d][c];
          if (wv >= 0 && (wv & (POOLROWS_NONPOS - 1)) == d) { a += g[r + 2 * PAD - d][c]; nonpos = wv & POOLROWS_NONPOS; }
        }
        o[c] = nonpos ? a * neg_slope : a;
      }
      const long e = base + (long)row * W;
      if constexpr (V == 4) *reinterpret_cast<float4*>(dh + e) = float4{o[0], o[1], o[2], o[3]};
      else if constexpr (V == 2) *reinterpret_cast<float2*>(dh + e) = float2{o[0], o[1]};
      else dh[e] = o[0];
    }
  }
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void axpy_kernel(float alpha, const float* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] += alpha * x[i];
}

__global__ __launch_bounds__(256) void scale_by_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       float* __restrict__ y, long n) {
  const float a = g[0];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = a * x[i];
}

__global__ __launch_bounds__(256) void scale_kernel(float alpha, float* __restrict__ x, long n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= alpha;
}

// y[b][c][r] = x[b][r][c] (+ pe)   x: (B,R,Cc) -> y: (B,Cc,R); LDS 32x33 tile transpose.
// pe_mode 0: none; 1: pe indexed like the OUTPUT's (row=c? no) -- see host wrapper.
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                        float* __restrict__ y, int R, int Cc, int pe_mode) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  const float* xb = x + (long)b * R * Cc;
  float* yb = y + (long)b * R * Cc;
  // the four loads of a thread are issued together, from clamped positions (a guarded load each is a branch and a
  // wait each); positions outside the matrix are never written back
  float v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int r = min(r0 + ty + 8 * u, R - 1), c = min(c0 + tx, Cc - 1);
    v[u] = xb[(long)r * Cc + c];
  }
  if (pe_mode == 2) {                                    // pe laid out like the input (R,Cc)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int r = min(r0 + ty + 8 * u, R - 1), c = min(c0 + tx, Cc - 1);
      v[u] += pe[(long)r * Cc + c];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) tile[ty + 8 * u][tx] = v[u];
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (r < R && c < Cc) {
      float v = tile[tx][k];
      if (pe_mode == 1) v += pe[(long)c * R + r];       // pe laid out like the output (Cc,R)
      yb[(long)c * R + r] = v;
    }
  }
}

// out[c] = sum_{b,i} x[b][c][i]   grid (C): one block per channel (bias gradients; small C, long rows)
__global__ __launch_bounds__(256) void channel_sum_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int C,
                                                          int HW) {
  const int c = blockIdx.x;
  const long per = (long)B * HW;
  double acc = 0.0;
  float s = 0.f;
  int cnt = 0;
  for (long i = threadIdx.x; i < per; i += 256) {
    const int b = (int)(i / HW);
    const int r = (int)(i - (long)b * HW);
    s += x[((long)b * C + c) * HW + r];
    if (++cnt == 64) { acc += s; s = 0.f; cnt = 0; }
  }
  acc += s;
  acc = mpa_wave_sum_d(acc);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// out[n] += sum_rows x[row][n]; grid (ceil(N/64), row splits); block 256 = 4 row-groups x 64 columns; float atomics
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* __restrict__ out, long rows, int N) {
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  float s = 0.f;
  if (col < N) {
    // four independent chains keep four loads in flight per thread
    const long step = (long)gridDim.y * 4;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    long r = (long)blockIdx.y * 4 + g;
    for (; r + 3 * step < rows; r += 4 * step) {
      a0 += x[r * N + col];
      a1 += x[(r + step) * N + col];
      a2 += x[(r + 2 * step) * N + col];
      a3 += x[(r + 3 * step) * N + col];
    }
    for (; r < rows; r += step) a0 += x[r * N + col];
    s = (a0 + a1) + (a2 + a3);
  }
  __shared__ float sh[4][64];
  sh[g][threadIdx.x & 63] = s;
  __syncthreads();
  if (g == 0 && col < N)
    atomicAdd(&out[col], sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void add_rows_bcast_kernel(const float* __restrict__ x, const float* __restrict__ pe,
                                                             float* __restrict__ y, long n, long SE) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) y[i] = x[i] + pe[i % SE];
}

// ------------------------------------------------------------------ losses
__global__ __launch_bounds__(256) void bce_fwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                      float* __restrict__ loss, long n, float inv_n) {
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pi = p[i], yi = y[i];
    const float lp = fmaxf(logf(pi), -100.f), lq = fmaxf(logf(1.f - pi), -100.f);
    s -= yi * lp + (1.f - yi) * lq;
  }
  s = mpa_wave_sum(s);
  __shared__ float sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss, (sh[0] + sh[1] + sh[2] + sh[3]) * inv_n);
}

// d/dp of mean BCE as ATen's binary_cross_entropy_backward: (p - y) / max((1-p)*p, 1e-12) * g / n
__global__ __launch_bounds__(256) void bce_bwd_kernel(const float* __restrict__ p, const float* __restrict__ y,
                                                      float* __restrict__ dp, long n, const float* __restrict__ g,
                                                      float inv_n) {
  const float gscale = (g ? g[0] : 1.f) * inv_n;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float pi = p[i];
    dp[i] = (pi - y[i]) / fmaxf((1.f - pi) * pi, 1e-12f) * gscale;
  }
}

// one wave per row; loss += scale * mean(lse - logit[target]); dlogits = scale/B * (softmax - onehot)
__global__ __launch_bounds__(64) void ce_kernel(const float* __restrict__ logits, const int64_t* __restrict__ target,
                                                float* __restrict__ loss, float* __restrict__ dlogits, int B, int K,
                                                float scale) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* row = logits + (long)b * K;
  float m = -INFINITY;
  for (int k = lane; k < K; k += 64) m = fmaxf(m, row[k]);
  m = mpa_wave_max(m);
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += expf(row[k] - m);
  s = mpa_wave_sum(s);
  const float lse = m + logf(s);
  const int t = (int)target[b];
  const float w = scale / (float)B;
  if (dlogits)
    for (int k = lane; k < K; k += 64) dlogits[(long)b * K + k] = w * (expf(row[k] - lse) - (k == t ? 1.f : 0.f));
  if (lane == 0) atomicAdd(loss, w * (lse - row[t]));
}

// ------------------------------------------------------------------ LSTM cell
// acts (B,4H) receives sigmoid(i), sigmoid(f), tanh(g), sigmoid(o)
__global__ __launch_bounds__(256) void lstm_cell_fwd_kernel(const float* __restrict__ gates, long g_stride,
                                                            const float* __restrict__ c_prev, float* __restrict__ c,
                                                            float* __restrict__ h, long h_stride, float* __restrict__ acts,
                                                            int B, int H) {
  const long n = (long)B * H;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int b = (int)(i / H), j = (int)(i - (long)b * H);
    const float* g = gates + (long)b * g_stride;
    const float ig = 1.f / (1.f + expf(-g[j]));
    const float fg = 1.f / (1.f + expf(-g[H + j]));
    const float gg = tanhf(g[2 * H + j]);
    const float og = 1.f / (1.f + expf(-g[3 * H + j]));
    const float cp = c_prev ? c_prev[i] : 0.f;
    const float cn = fg * cp + ig * gg;
    c[i] = cn;
    h[(long)b * h_stride + j] = og * tanhf(cn);
    float* a = acts + (long)b * 4 * H;
    a[j] = ig; a[H + j] = fg; a[2 * H + j] = gg; a[3 * H + j] = og;
  }
}

__global__ __launch_bounds__(256) void lstm_cell_bwd_kernel(const float* __restrict__ dh, long dh_stride,
                                                            const float* __restrict__ dh_rec,
                                                            const float* __restrict__ dc_next, const float* __restrict__ acts,
                                                            const float* __restrict__ c_prev, const float* __restrict__ c,
                                                            float* __restrict__ dgates, long dg_stride,
                                                            float* __restrict__ dc_prev, int B, int H) {
  const long n = (long)B * H;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int b = (int)(i / H), j = (int)(i - (long)b * H);
    const float* a = acts + (long)b * 4 * H;
    const float ig = a[j], fg = a[H + j], gg = a[2 * H + j], og = a[3 * H + j];
    const float tc = tanhf(c[i]);
    const float dht = dh[(long)b * dh_stride + j] + (dh_rec ? dh_rec[i] : 0.f);
    float dct = dht * og * (1.f - tc * tc) + (dc_next ? dc_next[i] : 0.f);
    const float cp = c_prev ? c_prev[i] : 0.f;
    float* dg = dgates + (long)b * dg_stride;
    dg[j] = dct * gg * ig * (1.f - ig);
    dg[H + j] = dct * cp * fg * (1.f - fg);
    dg[2 * H + j] = dct * ig * (1.f - gg * gg);
    dg[3 * H + j] = dht * tc * og * (1.f - og);
    dc_prev[i] = dct * fg;
  }
}

// pointer table upload without a memcpy: up to 64 pointers travel as kernel arguments (copied at launch time, so the
// host array may be reused immediately, and a captured graph replays them as constants)
struct PtrChunk { const void* p[64]; };
__global__ void store_ptrs_kernel(const void** __restrict__ table, PtrChunk c, int n) {
  if ((int)threadIdx.x < n) table[threadIdx.x] = c.p[threadIdx.x];
}

// multi-tensor copy: up to 32 tensors into one flat buffer per launch (gradient buckets of the data-parallel averager)
struct GatherArgs {
  const float* src[32];
  long off[32];
  long n[32];
};
__global__ __launch_bounds__(256) void gather_copy_kernel(float* __restrict__ dst, GatherArgs a) {
  const int t = blockIdx.y;
  const float* __restrict__ s = a.src[t];
  float* __restrict__ d = dst + a.off[t];
  const long n = a.n[t];
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) d[i] = s[i];
}

__global__ void u64_add_kernel(uint64_t* __restrict__ p, uint64_t delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += delta;
}

// ------------------------------------------------------------------ AdamW (multi-tensor)
// hyper (device, float64): [0] learning rate, [1] number of steps taken, [2] 1 - beta1^step, [3] sqrt(1 - beta2^step).
// Device-resident so that the step can sit inside a captured HIP graph: ReduceLROnPlateau writes hyper[0],
// adamw_advance_kernel counts the step and refreshes the bias corrections (in double, as torch.optim.AdamW does on the
// host) right before the update kernel reads them.
__global__ void adamw_advance_kernel(double* __restrict__ hyper, double beta1, double beta2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double step = hyper[1] + 1.0;
    hyper[1] = step;
    hyper[2] = 1.0 - pow(beta1, step);
    hyper[3] = sqrt(1.0 - pow(beta2, step));
  }
}

__global__ __launch_bounds__(256) void adamw_kernel(float* const* __restrict__ params, const float* const* __restrict__ grads,
                                                    float* const* __restrict__ m_, float* const* __restrict__ v_,
                                                    const int64_t* __restrict__ sizes, const double* __restrict__ hyper,
                                                    float b1, float b2, float eps, float wd) {
  const int t = blockIdx.y;
  const long n = sizes[t];
  float* p = params[t];
  const float* g = grads[t];
  float* m = m_[t];
  float* v = v_[t];
  if (!g) return;
  const float lr = (float)hyper[0], bc1 = (float)hyper[2], bc2_sqrt = (float)hyper[3];
  const float step_size = lr / bc1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float gi = g[i];
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
  }
}

template <int KH, int R>
int poolrows_fwd_launch(const float* h, const float* residual, float* out, int8_t* which, long planes, int H, int W, float p,
                        const uint64_t* rng_state, uint64_t offset, hipStream_t s) {
  const float scale = 1.f / (1.f - p);
  const uintptr_t al = (uintptr_t)h | (uintptr_t)out | (uintptr_t)residual | (uintptr_t)which;
  constexpr int VMAX = KH == 3 ? 4 : 2;      // the 13-row window keeps R + 12 rows per column in registers
  const int V = (VMAX == 4 && W % 4 == 0 && (al & 15) == 0) ? 4 : ((W % 2 == 0 && (al & 7) == 0) ? 2 : 1);
  const long items = planes * mpa_cdiv(H, R) * (W / V);
  if (V == 4) {
    if constexpr (VMAX == 4)
      MPA_LAUNCH((poolrows_drop_add_fwd_kernel<KH, R, 4>), dim3(blocks_for(items)), dim3(256), 0, s, h, residual, out, which, planes,
                 H, W, p, scale, rng_state, offset);
  } else if (V == 2) {
    MPA_LAUNCH((poolrows_drop_add_fwd_kernel<KH, R, 2>), dim3(blocks_for(items)), dim3(256), 0, s, h, residual, out, which, planes,
               H, W, p, scale, rng_state, offset);
  } else {
    MPA_LAUNCH((poolrows_drop_add_fwd_kernel<KH, R, 1>), dim3(blocks_for(items)), dim3(256), 0, s, h, residual, out, which, planes,
               H, W, p, scale, rng_state, offset);
  }
  return mpa_launch_status();
}
template <int KH, int R>
int poolrows_bwd_launch(const float* dout, const int8_t* which, float* dh, long planes, int H, int W, float p,
                        const uint64_t* rng_state, uint64_t offset, float neg_slope, hipStream_t s) {
  const float scale = 1.f / (1.f - p);
  const uintptr_t al = (uintptr_t)dout | (uintptr_t)dh | (uintptr_t)which;
  constexpr int VMAX = KH == 3 ? 4 : 2;
  const int V = (VMAX == 4 && W % 4 == 0 && (al & 15) == 0) ? 4 : ((W % 2 == 0 && (al & 7) == 0) ? 2 : 1);
  const long items = planes * mpa_cdiv(H, R) * (W / V);
  if (V == 4) {
    if constexpr (VMAX == 4)
      MPA_LAUNCH((poolrows_drop_bwd_kernel<KH, R, 4>), dim3(blocks_for(items)), dim3(256), 0, s, dout, which, dh, planes, H, W, p,
                 scale, rng_state, offset, neg_slope);
  } else if (V == 2) {
    MPA_LAUNCH((poolrows_drop_bwd_kernel<KH, R, 2>), dim3(blocks_for(items)), dim3(256), 0, s, dout, which, dh, planes, H, W, p,
               scale, rng_state, offset, neg_slope);
  } else {
    MPA_LAUNCH((poolrows_drop_bwd_kernel<KH, R, 1>), dim3(blocks_for(items)), dim3(256), 0, s, dout, which, dh, planes, H, W, p,
               scale, rng_state, offset, neg_slope);
  }
  return mpa_launch_status();
}

// nn.LogSoftmax(dim=1) over the channels of cat([a, b], dim=3) -- the output stage of basic_cnn_segm_logsoftmax /
// basic_cnn_segm_blank_logsoftmax (basic_cnns.py:254-255, 331-338).  a (B,C,R,Wa), b (B,C,R,Wb) or absent; one thread per
// (b, r, w) walks the C channels (a handful) three times: max, sum of exponentials, output.
__global__ void logsoftmax_cat_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long n,
                                          int C, int R, int Wa, int Wb) {
  const int W = Wa + Wb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long t = i / W;
    const int r = (int)(t % R);
    const long bb = t / R;
    const bool ina = w < Wa;
    const float* src = ina ? a + (bb * C * R + r) * (long)Wa + w : b + (bb * C * R + r) * (long)Wb + (w - Wa);
    const long cs = (long)R * (ina ? Wa : Wb);
    float m = -INFINITY;
    for (int c = 0; c < C; ++c) m = fmaxf(m, src[c * cs]);
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(src[c * cs] - m);
    const float lse = m + logf(sum);
    float* dst = y + (bb * C * R + r) * (long)W + w;
    for (int c = 0; c < C; ++c) dst[(long)c * R * W] = src[c * cs] - lse;
  }
}

// da / db = dy - exp(y) * sum_c dy
__global__ void logsoftmax_cat_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ da,
                                          float* __restrict__ db, long n, int C, int R, int Wa, int Wb) {
  const int W = Wa + Wb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int w = (int)(i % W);
    const long t = i / W;
    const int r = (int)(t % R);
    const long bb = t / R;
    const long o = (bb * C * R + r) * (long)W + w, ys = (long)R * W;
    float g = 0.f;
    for (int c = 0; c < C; ++c) g += dy[o + c * ys];
    const bool ina = w < Wa;
    float* dst = ina ? da + (bb * C * R + r) * (long)Wa + w : db + (bb * C * R + r) * (long)Wb + (w - Wa);
    const long cs = (long)R * (ina ? Wa : Wb);
    for (int c = 0; c < C; ++c) dst[c * cs] = dy[o + c * ys] - expf(y[o + c * ys]) * g;
  }
}

}  // namespace

extern "C" {

const char* mpa_strerror(int code) {
  switch (code) {
    case MPA_OK: return "ok";
    case MPA_ERR_ARG: return "invalid argument";
    case MPA_ERR_LAUNCH: return "kernel launch failed";
    case MPA_ERR_UNSUPPORTED: return "unsupported configuration";
    case MPA_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}
int mpa_version(void) { return 1; }

int mpa_diag_reload(void) {
  mpa_diag_mutable() = mpa_diag_read();
#ifdef MPA_DIAG
  return 1;
#else
  return 0;
#endif
}
int mpa_logsoftmax_cat_fwd(const float* a, const float* b, float* y, int B, int C, int R, int Wa, int Wb, void* stream) {
  if (!a || !y || B <= 0 || C <= 0 || R <= 0 || Wa <= 0 || Wb < 0 || (Wb > 0 && !b)) return MPA_ERR_ARG;
  const long n = (long)B * R * (Wa + Wb);
  MPA_LAUNCH(logsoftmax_cat_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, n, C, R, Wa, Wb);
  return mpa_launch_status();
}
int mpa_logsoftmax_cat_bwd(const float* dy, const float* y, float* da, float* db, int B, int C, int R, int Wa, int Wb,
                           void* stream) {
  if (!dy || !y || !da || B <= 0 || C <= 0 || R <= 0 || Wa <= 0 || Wb < 0 || (Wb > 0 && !db)) return MPA_ERR_ARG;
  const long n = (long)B * R * (Wa + Wb);
  MPA_LAUNCH(logsoftmax_cat_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, da, db, n, C, R, Wa, Wb);
  return mpa_launch_status();
}


int mpa_act_fwd(const float* x, float* y, int64_t n, int act, float slope, void* stream) {
  if (!x || !y) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  MPA_LAUNCH(act_fwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, act, slope);
  return mpa_launch_status();
}
int mpa_act_bwd(const float* dy, const float* x, float* dx, int64_t n, int act, float slope, void* stream) {
  if (!dy || !x || !dx) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  MPA_LAUNCH(act_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, dy, x, dx, (long)n, act, slope);
  return mpa_launch_status();
}
int mpa_dropout(const float* x, float* y, int64_t n, float p, const uint64_t* rng_state, uint64_t offset, void* stream) {
  if (!x || !y || !rng_state || p < 0.f || p >= 1.f) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  MPA_LAUNCH(dropout_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, (long)n, p, 1.f / (1.f - p),
                     rng_state, offset);
  return mpa_launch_status();
}
int mpa_poolrows_dropout_add_fwd(const float* h, const float* residual, float* out, int8_t* which, int64_t planes, int H,
                                 int W, int kh, float p, const uint64_t* rng_state, uint64_t offset, void* stream) {
  if (!h || !out || planes < 0 || H <= 0 || W <= 0 || p < 0.f || p >= 1.f || (p > 0.f && !rng_state)) return MPA_ERR_ARG;
  if (kh != 3 && kh != 13) return MPA_ERR_UNSUPPORTED;
  if (planes == 0) return MPA_OK;
  if (kh == 3) return poolrows_fwd_launch<3, 15>(h, residual, out, which, (long)planes, H, W, p, rng_state, offset, (hipStream_t)stream);
  return poolrows_fwd_launch<13, 12>(h, residual, out, which, (long)planes, H, W, p, rng_state, offset, (hipStream_t)stream);
}
int mpa_poolrows_dropout_act_bwd(const float* dout, const int8_t* which, float* dh, int64_t planes, int H, int W, int kh, float p,
                                 const uint64_t* rng_state, uint64_t offset, float neg_slope, void* stream) {
  if (!dout || !which || !dh || planes < 0 || H <= 0 || W <= 0 || p < 0.f || p >= 1.f || (p > 0.f && !rng_state))
    return MPA_ERR_ARG;
  if (kh != 3 && kh != 13) return MPA_ERR_UNSUPPORTED;
  if (planes == 0) return MPA_OK;
  if (kh == 3)
    return poolrows_bwd_launch<3, 15>(dout, which, dh, (long)planes, H, W, p, rng_state, offset, neg_slope, (hipStream_t)stream);
  return poolrows_bwd_launch<13, 12>(dout, which, dh, (long)planes, H, W, p, rng_state, offset, neg_slope, (hipStream_t)stream);
}
int mpa_poolrows_dropout_bwd(const float* dout, const int8_t* which, float* dh, int64_t planes, int H, int W, int kh, float p,
                             const uint64_t* rng_state, uint64_t offset, void* stream) {
  return mpa_poolrows_dropout_act_bwd(dout, which, dh, planes, H, W, kh, p, rng_state, offset, 1.f, stream);
}
int mpa_store_ptrs(const void** table, const void* const* host_ptrs, int n, void* stream) {
  if (!table || !host_ptrs || n < 0) return MPA_ERR_ARG;
  for (int o = 0; o < n; o += 64) {
    PtrChunk c{};
    const int m = n - o < 64 ? n - o : 64;
    for (int i = 0; i < m; ++i) c.p[i] = host_ptrs[o + i];
    MPA_LAUNCH(store_ptrs_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, table + o, c, m);
    const int rc = mpa_launch_status();
    if (rc) return rc;
  }
  return MPA_OK;
}
int mpa_gather_copy(float* dst, const float* const* srcs, const int64_t* dst_offsets, const int64_t* sizes, int n,
                    void* stream) {
  if (!dst || !srcs || !dst_offsets || !sizes || n < 0) return MPA_ERR_ARG;
  for (int o = 0; o < n; o += 32) {
    GatherArgs a{};
    const int m = n - o < 32 ? n - o : 32;
    long mx = 1;
    for (int i = 0; i < m; ++i) {
      if (!srcs[o + i] || sizes[o + i] < 0) return MPA_ERR_ARG;
      a.src[i] = srcs[o + i]; a.off[i] = dst_offsets[o + i]; a.n[i] = sizes[o + i];
      mx = std::max<long>(mx, sizes[o + i]);
    }
    dim3 grid((unsigned)std::max<long>(1, std::min<long>(mpa_cdiv(mx, 1024), 256)), (unsigned)m);
    MPA_LAUNCH(gather_copy_kernel, grid, dim3(256), 0, (hipStream_t)stream, dst, a);
    const int rc = mpa_launch_status();
    if (rc) return rc;
  }
  return MPA_OK;
}
int mpa_u64_add(uint64_t* counter, uint64_t delta, void* stream) {
  if (!counter) return MPA_ERR_ARG;
  MPA_LAUNCH(u64_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, delta);
  return mpa_launch_status();
}
int mpa_add(const float* a, const float* b, float* y, int64_t n, void* stream) {
  if (!a || !b || !y) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  MPA_LAUNCH(add_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, y, (long)n);
  return mpa_launch_status();
}
int mpa_axpy(float alpha, const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  MPA_LAUNCH(axpy_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, alpha, x, y, (long)n);
  return mpa_launch_status();
}
int mpa_scale_by(const float* x, const float* g, float* y, int64_t n, void* stream) {
  if (!x || !g || !y) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  MPA_LAUNCH(scale_by_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, g, y, (long)n);
  return mpa_launch_status();
}
int mpa_scale(float alpha, float* x, int64_t n, void* stream) {
  if (!x) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  MPA_LAUNCH(scale_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, alpha, x, (long)n);
  return mpa_launch_status();
}
// x (B,R,Cc) -> y (B,Cc,R); pe_mode 0 none, 1 add pe (Cc,R) to the output, 2 add pe (R,Cc) to the input
int mpa_transpose_add(const float* x, const float* pe, float* y, int B, int R, int Cc, int pe_mode, void* stream) {
  if (!x || !y || (pe_mode && !pe)) return MPA_ERR_ARG;
  dim3 grid((unsigned)mpa_cdiv(Cc, 32), (unsigned)mpa_cdiv(R, 32), (unsigned)B);
  MPA_LAUNCH(transpose_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, pe, y, R, Cc, pe_mode);
  return mpa_launch_status();
}
int mpa_channel_sum(const float* x, float* out, int B, int C, int HW, void* stream) {
  if (!x || !out) return MPA_ERR_ARG;
  MPA_LAUNCH(channel_sum_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, out, B, C, HW);
  return mpa_launch_status();
}
int mpa_colsum(const float* x, float* out, int64_t rows, int N, int accumulate, void* stream) {
  if (!x || !out) return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!accumulate && mpa_zero_async(out, sizeof(float) * (size_t)N, s) != MPA_OK) return MPA_ERR_LAUNCH;
  const long colblocks = mpa_cdiv(N, 64);
  long splits = std::max<long>(1, std::min<long>(mpa_cdiv(rows, 64), mpa_cdiv(1024, colblocks)));
  MPA_LAUNCH(colsum_kernel, dim3((unsigned)colblocks, (unsigned)splits), dim3(256), 0, s, x, out, (long)rows, N);
  return mpa_launch_status();
}
int mpa_add_rows_bcast(const float* x, const float* pe, float* y, int B, int64_t SE, void* stream) {
  if (!x || !pe || !y) return MPA_ERR_ARG;
  const long n = (long)B * SE;
  MPA_LAUNCH(add_rows_bcast_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, pe, y, n, (long)SE);
  return mpa_launch_status();
}

int mpa_bce_fwd(const float* p, const float* y, float* loss_out, int64_t n, void* stream) {
  if (!p || !y || !loss_out || n <= 0) return MPA_ERR_ARG;
  if (mpa_zero_async(loss_out, sizeof(float), (hipStream_t)stream) != MPA_OK) return MPA_ERR_LAUNCH;   // the blocks add into it
  MPA_LAUNCH(bce_fwd_kernel, dim3(blocks_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, p, y, loss_out, (long)n,
                     1.0f / (float)n);
  return mpa_launch_status();
}
int mpa_bce_bwd(const float* p, const float* y, float* dp, int64_t n, const float* g, void* stream) {
  if (!p || !y || !dp || n <= 0) return MPA_ERR_ARG;
  MPA_LAUNCH(bce_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, p, y, dp, (long)n, g,
                     1.0f / (float)n);
  return mpa_launch_status();
}
int mpa_ce_fwd_bwd(const float* logits, const int64_t* target, float* loss_out, float* dlogits, int B, int K, float scale,
                   void* stream) {
  if (!logits || !target || !loss_out || B <= 0) return MPA_ERR_ARG;
  if (mpa_zero_async(loss_out, sizeof(float), (hipStream_t)stream) != MPA_OK) return MPA_ERR_LAUNCH;   // the rows add into it
  MPA_LAUNCH(ce_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, logits, target, loss_out, dlogits, B, K, scale);
  return mpa_launch_status();
}

int mpa_lstm_cell_fwd(const float* gates, int64_t g_stride, const float* c_prev, float* c, float* h, int64_t h_stride,
                      float* acts, int B, int H, void* stream) {
  if (!gates || !c || !h || !acts) return MPA_ERR_ARG;
  MPA_LAUNCH(lstm_cell_fwd_kernel, dim3(blocks_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, gates,
                     (long)g_stride, c_prev, c, h, (long)h_stride, acts, B, H);
  return mpa_launch_status();
}
int mpa_lstm_cell_bwd(const float* dh, int64_t dh_stride, const float* dh_rec, const float* dc_next, const float* acts,
                      const float* c_prev, const float* c, float* dgates, int64_t dg_stride, float* dc_prev, int B, int H,
                      void* stream) {
  if (!dh || !acts || !c || !dgates || !dc_prev) return MPA_ERR_ARG;
  MPA_LAUNCH(lstm_cell_bwd_kernel, dim3(blocks_for((long)B * H)), dim3(256), 0, (hipStream_t)stream, dh,
                     (long)dh_stride, dh_rec, dc_next, acts, c_prev, c, dgates, (long)dg_stride, dc_prev, B, H);
  return mpa_launch_status();
}

int mpa_adamw_step(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                   const int64_t* sizes, int ntensors, int64_t max_size, double* hyper, double beta1, double beta2, double eps,
                   double weight_decay, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !sizes || !hyper || ntensors <= 0 || max_size <= 0) return MPA_ERR_ARG;
  MPA_LAUNCH(adamw_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper, beta1, beta2);
  int rc = mpa_launch_status();
  if (rc) return rc;
  dim3 grid((unsigned)std::max<long>(1, std::min<long>(mpa_cdiv(max_size, 1024), 512)), (unsigned)ntensors);
  MPA_LAUNCH(adamw_kernel, grid, dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, sizes,
             (const double*)hyper, (float)beta1, (float)beta2, (float)eps, (float)weight_decay);
  return mpa_launch_status();
}

}  // extern "C"
