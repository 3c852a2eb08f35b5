// MaxPool2d (fwd/bwd with first-max index) and the U-Net decoder's bilinear x2 upsample + pad + concat.
#include "mpa_common.h"
#include <algorithm>

namespace {

__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          int32_t* __restrict__ idx, long planes, int H, int W, int OH,
                                                          int OW, int kh, int kw, int sh, int sw, int ph, int pw) {
  const long total = planes * OH * OW;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % OW);
    const long r = i / OW;
    const int oy = (int)(r % OH);
    const long pl = r / OH;
    const float* xp = x + pl * H * W;
    const int y0 = oy * sh - ph, x0 = ox * sw - pw;
    float best = -INFINITY;
    int bi = -1;
    for (int dy = 0; dy < kh; ++dy) {
      const int iy = y0 + dy;
      if (iy < 0 || iy >= H) continue;
      for (int dx = 0; dx < kw; ++dx) {
        const int ix = x0 + dx;
        if (ix < 0 || ix >= W) continue;
        const float v = xp[iy * W + ix];
        if (bi < 0 || v > best || v != v) { best = v; bi = iy * W + ix; }   // first maximum wins; NaN propagates
      }
    }
    y[i] = best;
    if (idx) idx[i] = bi;
  }
}

// gather form: every input element sums dy of the windows whose recorded argmax it is (deterministic, no atomics)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx,
                                                          float* __restrict__ dx, long planes, int H, int W, int OH,
                                                          int OW, int kh, int kw, int sh, int sw, int ph, int pw) {
  const long total = planes * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ix = (int)(i % W);
    const long r = i / W;
    const int iy = (int)(r % H);
    const long pl = r / H;
    const int me = iy * W + ix;
    // windows oy with oy*sh - ph <= iy <= oy*sh - ph + kh - 1
    int oy_lo = (iy + ph - kh + 1 + sh - 1);
    oy_lo = oy_lo <= 0 ? 0 : oy_lo / sh;
    int oy_hi = (iy + ph) / sh;
    if (oy_hi > OH - 1) oy_hi = OH - 1;
    int ox_lo = (ix + pw - kw + 1 + sw - 1);
    ox_lo = ox_lo <= 0 ? 0 : ox_lo / sw;
    int ox_hi = (ix + pw) / sw;
    if (ox_hi > OW - 1) ox_hi = OW - 1;
    float s = 0.f;
    const long ob = pl * OH * OW;
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const long o = ob + (long)oy * OW + ox;
        if (idx[o] == me) s += dy[o];
      }
    dx[i] = s;
  }
}

// align_corners=True source index in float32, as ATen's area_pixel_compute_source_index
__device__ __forceinline__ void bilin_src(int dst, int n_in, int n_out, int& i0, int& i1, float& l) {
  if (n_in <= 1 || n_out <= 1) { i0 = 0; i1 = 0; l = 0.f; return; }
  const float scale = (float)(n_in - 1) / (float)(n_out - 1);
  const float src = scale * (float)dst;
  i0 = (int)src;
  if (i0 > n_in - 1) i0 = n_in - 1;
  i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
  l = src - (float)i0;
}

__global__ __launch_bounds__(256) void upcat_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ skip,
                                                        float* __restrict__ out, int B, int C1, int H1, int W1, int Cs,
                                                        int Hs, int Ws) {
  const int Ct = Cs + C1;
  const long total = (long)B * Ct * Hs * Ws;
  const int UH = 2 * H1, UW = 2 * W1;
  const int padT = (Hs - UH) / 2, padL = (Ws - UW) / 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int x = (int)(i % Ws);
    long r = i / Ws;
    const int y = (int)(r % Hs);
    r /= Hs;
    const int c = (int)(r % Ct);
    const int b = (int)(r / Ct);
    float v;
    if (c < Cs) {
      v = skip[(((long)b * Cs + c) * Hs + y) * Ws + x];
    } else {
      const int uy = y - padT, ux = x - padL;
      if (uy < 0 || uy >= UH || ux < 0 || ux >= UW) {
        v = 0.f;
      } else {
        int y0, y1, x0, x1i;
        float ly, lx;
        bilin_src(uy, H1, UH, y0, y1, ly);
        bilin_src(ux, W1, UW, x0, x1i, lx);
        const float* p = x1 + ((long)b * C1 + (c - Cs)) * H1 * W1;
        const float hy = 1.f - ly, hx = 1.f - lx;
        v = hy * (hx * p[y0 * W1 + x0] + lx * p[y0 * W1 + x1i]) + ly * (hx * p[y1 * W1 + x0] + lx * p[y1 * W1 + x1i]);
      }
    }
    out[i] = v;
  }
}

// dskip = dout[:, :Cs]; dx1[i,j] = sum over upsampled positions that read (i,j)
__global__ __launch_bounds__(256) void upcat_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx1,
                                                        float* __restrict__ dskip, int B, int C1, int H1, int W1, int Cs,
                                                        int Hs, int Ws) {
  const int Ct = Cs + C1;
  const int UH = 2 * H1, UW = 2 * W1;
  const int padT = (Hs - UH) / 2, padL = (Ws - UW) / 2;
  const long nskip = (long)B * Cs * Hs * Ws, nx1 = (long)B * C1 * H1 * W1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nskip + nx1; i += (long)gridDim.x * 256) {
    if (i < nskip) {
      const long plane = Hs * (long)Ws;
      const long b = i / (Cs * plane), rem = i - b * (Cs * plane);
      dskip[i] = dout[b * Ct * plane + rem];
      continue;
    }
    const long k = i - nskip;
    const int j = (int)(k % W1);
    long r = k / W1;
    const int ii = (int)(r % H1);
    r /= H1;
    const int c = (int)(r % C1);
    const int b = (int)(r / C1);
    const float* dp = dout + (((long)b * Ct + Cs + c) * Hs) * Ws;
    float s = 0.f;
    const int uy_lo = max(0, 2 * ii - 2), uy_hi = min(UH - 1, 2 * ii + 3);
    const int ux_lo = max(0, 2 * j - 2), ux_hi = min(UW - 1, 2 * j + 3);
    for (int uy = uy_lo; uy <= uy_hi; ++uy) {
      int y0, y1;
      float ly;
      bilin_src(uy, H1, UH, y0, y1, ly);
      float wy = 0.f;
      if (y0 == ii) wy += 1.f - ly;
      if (y1 == ii) wy += ly;
      if (wy == 0.f) continue;
      const int oy = uy + padT;
      if (oy < 0 || oy >= Hs) continue;
      for (int ux = ux_lo; ux <= ux_hi; ++ux) {
        int x0, x1i;
        float lx;
        bilin_src(ux, W1, UW, x0, x1i, lx);
        float wx = 0.f;
        if (x0 == j) wx += 1.f - lx;
        if (x1i == j) wx += lx;
        if (wx == 0.f) continue;
        const int ox = ux + padL;
        if (ox < 0 || ox >= Ws) continue;
        s += wy * wx * dp[oy * Ws + ox];
      }
    }
    dx1[k] = s;
  }
}

inline unsigned blocks_for(long n) { return (unsigned)std::max<long>(1, std::min<long>(mpa_cdiv(n, 256), 1 << 16)); }

}  // namespace

extern "C" {

int mpa_maxpool2d_fwd(const float* x, float* y, int32_t* idx, int B, int C, int H, int W, int kh, int kw, int sh, int sw,
                      int ph, int pw, void* stream) {
  if (!x || !y) return MPA_ERR_ARG;
  const int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
  if (OH <= 0 || OW <= 0) return MPA_ERR_ARG;
  const long planes = (long)B * C;
  MPA_LAUNCH(maxpool_fwd_kernel, dim3(blocks_for(planes * OH * OW)), dim3(256), 0, (hipStream_t)stream, x, y, idx,
                     planes, H, W, OH, OW, kh, kw, sh, sw, ph, pw);
  return mpa_launch_status();
}

int mpa_maxpool2d_bwd(const float* dy, const int32_t* idx, float* dx, int B, int C, int H, int W, int kh, int kw, int sh,
                      int sw, int ph, int pw, void* stream) {
  if (!dy || !idx || !dx) return MPA_ERR_ARG;
  const int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
  const long planes = (long)B * C;
  MPA_LAUNCH(maxpool_bwd_kernel, dim3(blocks_for(planes * H * W)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx,
                     planes, H, W, OH, OW, kh, kw, sh, sw, ph, pw);
  return mpa_launch_status();
}

int mpa_upcat_fwd(const float* x1, const float* skip, float* out, int B, int C1, int H1, int W1, int Cs, int Hs, int Ws,
                  void* stream) {
  if (!x1 || !skip || !out || Hs < 2 * H1 || Ws < 2 * W1) return MPA_ERR_ARG;
  MPA_LAUNCH(upcat_fwd_kernel, dim3(blocks_for((long)B * (Cs + C1) * Hs * Ws)), dim3(256), 0, (hipStream_t)stream, x1,
                     skip, out, B, C1, H1, W1, Cs, Hs, Ws);
  return mpa_launch_status();
}

int mpa_upcat_bwd(const float* dout, float* dx1, float* dskip, int B, int C1, int H1, int W1, int Cs, int Hs, int Ws,
                  void* stream) {
  if (!dout || !dx1 || !dskip) return MPA_ERR_ARG;
  const long n = (long)B * Cs * Hs * Ws + (long)B * C1 * H1 * W1;
  MPA_LAUNCH(upcat_bwd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, dout, dx1, dskip, B, C1, H1, W1,
                     Cs, Hs, Ws);
  return mpa_launch_status();
}

}  // extern "C"
