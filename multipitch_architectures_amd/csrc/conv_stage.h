// LDS staging helpers shared by the fp32 convolution kernels (LDS-DMA stagers, the row-end edge fix): device code that
// conv_fwd.hip, conv_wgrad.hip and conv_wgrad15.hip include (internal linkage, one copy per translation unit).
#pragma once
#include "mpa_common.h"

namespace {

__device__ __forceinline__ void fast_divmod(int idx, int d, float inv, int& q, int& r) {
  q = (int)((float)idx * inv);
  r = idx - q * d;
  if (r < 0) { q -= 1; r += d; }
  else if (r >= d) { q += 1; r -= d; }
}

// Stage a (nch x nrows x ncols) window of an NCHW image plane set into LDS, zero-filling outside the image.
// dst[ch*chp + iy*lw + ix] = src[(c0+ch), y0+iy, x0+ix]
__device__ __forceinline__ void stage_window(float* __restrict__ dst, const float* __restrict__ src, int tid, int nch,
                                             int nrows, int ncols, int chp, int lw, int c0, int y0, int x0, int C, int H,
                                             int W, int xlim) {
  // element idx = tid + 256*k walks (ch, iy, ix) incrementally: no per-element division
  const int total = nch * nrows * ncols;
  int row, ix, ch, iy;
  fast_divmod(tid, ncols, 1.0f / (float)ncols, row, ix);
  fast_divmod(row, nrows, 1.0f / (float)nrows, ch, iy);
  int dq, dr;
  fast_divmod(256, ncols, 1.0f / (float)ncols, dq, dr);
  for (int base = tid; base < total; base += 256 * 4) {
    float v[4];
    int o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      v[u] = 0.f;
      o[u] = -1;
      if (base + u * 256 < total) {
        const int gc = c0 + ch, gy = y0 + iy, gx = x0 + ix;
        o[u] = ch * chp + iy * lw + ix;
        if (gc < C && gy >= 0 && gy < H && gx >= 0 && gx < xlim) v[u] = src[((long)gc * H + gy) * W + gx];
      }
      ix += dr;
      iy += dq;
      if (ix >= ncols) { ix -= ncols; iy += 1; }
      while (iy >= nrows) { iy -= nrows; ch += 1; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (o[u] >= 0) dst[o[u]] = v[u];
  }
}

// LDS-DMA staging (global_load_lds_dword): every LDS word of the image is fetched straight from global memory
// -- or from a zero word when it lies outside the tensor -- with no VGPR round trip, so a whole tile (~60 loads per
// lane) is in flight at once instead of being paid for in dependent batches.  The address arithmetic is branch-free
// and 32-bit: measured with s_memtime stamps, a branchy per-word decode made the *issue* of a tile's loads take as
// long as its MFMA loop.
__device__ __attribute__((aligned(16))) float mpa_zero_src[256];

__device__ __forceinline__ void glds_word(const float* src, float* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 4, 0, 0);
}

// image [nch][chp] whose first nrows*lw words per channel hold rows of pitch lw; padded to `total64` words.
// Offsets are 32-bit (one image's plane set < 2^31 elements).
__device__ __forceinline__ void glds_stage_x(float* __restrict__ dst, const float* __restrict__ src, int lane, int wave,
                                             int nch, int nrows, int ncols, int lw, int chp, int total64, int c0, int y0,
                                             int x0, int C, int H, int W) {
  int ch, r;
  fast_divmod(wave * 64 + lane, chp, 1.0f / (float)chp, ch, r);
  const float inv_lw = 1.0f / (float)lw;
  const int used = nrows * lw;
  const int cmax = min(nch, C - c0);          // channels of this window that exist
  const float* zsrc = &mpa_zero_src[lane];
  const int HW = H * W;
  for (int base = wave * 64; base < total64; base += 256) {
    int iy, ix;
    fast_divmod(r, lw, inv_lw, iy, ix);
    const int gy = y0 + iy, gx = x0 + ix;
    const int ok = (int)(ch < cmax) & (int)(r < used) & (int)(ix < ncols) & (int)((unsigned)gy < (unsigned)H) &
                   (int)((unsigned)gx < (unsigned)W);
    const int off = (c0 + ch) * HW + gy * W + gx;
    const float* ptr = ok ? src + off : zsrc;
    glds_word(ptr, dst + base);
    r += 256;
    const int wrap = r >= chp;
    r -= wrap ? chp : 0;
    ch += wrap;
    if (r >= chp) {                              // tiny images only (chp < 256)
      while (r >= chp) { r -= chp; ch += 1; }
    }
  }
}

// image [nco][dcp] with the first th*dp words of a row holding (py, px); padded to `total64`
__device__ __forceinline__ void glds_stage_dy(float* __restrict__ dst, const float* __restrict__ src, int lane, int wave,
                                              int nco, int th, int dp, int dcp, int total64, int c0, int y0, int x0, int C,
                                              int OH, int OW, int xlim) {
  int co, r;
  fast_divmod(wave * 64 + lane, dcp, 1.0f / (float)dcp, co, r);
  const float inv_dp = 1.0f / (float)dp;
  const float* zsrc = &mpa_zero_src[lane];
  const int npx = th * dp;
  const int cmax = min(nco, C - c0);
  const int plane = OH * OW;
  for (int base = wave * 64; base < total64; base += 256) {
    int py, px;
    fast_divmod(r, dp, inv_dp, py, px);
    const int oy = y0 + py, ox = x0 + px;
    const int ok = (int)(co < cmax) & (int)(r < npx) & (int)(oy < OH) & (int)(ox < xlim);
    const int off = (c0 + co) * plane + oy * OW + ox;
    const float* ptr = ok ? src + off : zsrc;
    glds_word(ptr, dst + base);
    r += 256;
    const int wrap = r >= dcp;
    r -= wrap ? dcp : 0;
    co += wrap;
    if (r >= dcp) {
      while (r >= dcp) { r -= dcp; co += 1; }
    }
  }
}

// 16-byte LDS-DMA variants (4x fewer wave-instructions; the LDS-DMA path costs ~60-110 cycles per instruction per CU
// whatever its width -- measured).  Require: W % 4 == 0, window x origin x0a % 4 == 0 (so every float4 is entirely
// inside or outside the tensor), lw % 4 == 0, chp % 4 == 0.
__device__ __forceinline__ void glds_quad(const float* src, float* lds_dst_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_dst_wave_base, 16, 0, 0);
}

// EF (edge fix): W % 4 != 0 -- see edge_fix_*; compiled separately so that the common aligned case keeps its leaner loop
template <bool EF = false>
__device__ __forceinline__ void glds_stage_x16(float* __restrict__ dst, const float* __restrict__ src, int lane, int wave,
                                               int nch, int nrows, int lw, int chp, int total64, int c0, int y0, int x0a,
                                               int C, int H, int W) {
  const int lw4 = lw >> 2, chp4 = chp >> 2, total4 = total64 >> 2;     // everything in float4 units
  int ch, r;
  fast_divmod(wave * 64 + lane, chp4, 1.0f / (float)chp4, ch, r);
  const float inv = 1.0f / (float)lw4;
  const int used4 = nrows * lw4;
  const int cmax = min(nch, C - c0);
  const float* zsrc = &mpa_zero_src[(lane & 15) * 4];
  const int HW = H * W;
  for (int base = wave * 64; base < total4; base += 256) {
    int iy, q;
    fast_divmod(r, lw4, inv, iy, q);
    const int gy = y0 + iy, gx = x0a + 4 * q;
    if constexpr (EF) {
      // x0a is a multiple of 4 whenever it is negative, so gx >= 0 covers the left edge.  A quad that straddles the
      // right edge of its row is *not written here at all* (no DMA, so nothing can land late on top of it):
      // edge_fix_* owns it.
      const int rowok = (int)(ch < cmax) & (int)(r < used4) & (int)((unsigned)gy < (unsigned)H) & (int)(gx >= 0);
      const int full = rowok & (int)(gx + 3 < W);
      const int part = rowok & (int)(gx < W) & (int)(gx + 3 >= W);
      const int off = (c0 + ch) * HW + gy * W + gx;
      if (base + lane < total4 && !part) glds_quad(full ? src + off : zsrc, dst + (long)base * 4);
    } else {
      const int ok = (int)(ch < cmax) & (int)(r < used4) & (int)((unsigned)gy < (unsigned)H) & (int)((unsigned)gx < (unsigned)W);
      const int off = (c0 + ch) * HW + gy * W + gx;
      if (base + lane < total4)       // the image is a multiple of 16 float4, not of 64: never spill into the next region
        glds_quad(ok ? src + off : zsrc, dst + (long)base * 4);
    }
    r += 256;
    while (r >= chp4) { r -= chp4; ch += 1; }
  }
}

// dY image [nco][dcp]: first th*dp words per cout are rows (py) of dp words; dp % 4 == 0, dcp % 4 == 0.  The global side
// needs no alignment (16-byte LDS-DMA accepts any 4-byte aligned address); a quad straddling xlim is zero-filled here and
// completed by edge_fix_*
template <bool EF = false>
__device__ __forceinline__ void glds_stage_dy16(float* __restrict__ dst, const float* __restrict__ src, int lane, int wave,
                                                int nco, int th, int dp, int dcp, int total64, int c0, int y0, int x0, int C,
                                                int OH, int OW, int xlim) {
  const int dp4 = dp >> 2, dcp4 = dcp >> 2, total4 = total64 >> 2;
  int co, r;
  fast_divmod(wave * 64 + lane, dcp4, 1.0f / (float)dcp4, co, r);
  const float inv = 1.0f / (float)dp4;
  const float* zsrc = &mpa_zero_src[(lane & 15) * 4];
  const int n4 = th * dp4;
  const int cmax = min(nco, C - c0);
  const int plane = OH * OW;
  for (int base = wave * 64; base < total4; base += 256) {
    int py, q;
    fast_divmod(r, dp4, inv, py, q);
    const int oy = y0 + py, ox = x0 + 4 * q;
    const int rowok = (int)(co < cmax) & (int)(r < n4) & (int)(oy < OH);
    const int ok = rowok & (int)(ox + 3 < xlim);
    const int part = EF ? (rowok & (int)(ox < xlim) & (int)(ox + 3 >= xlim)) : 0;   // left to edge_fix_* (see there)
    const int off = (c0 + co) * plane + oy * OW + ox;
    if (base + lane < total4 && !part)
      glds_quad(ok ? src + off : zsrc, dst + (long)base * 4);
    r += 256;
    while (r >= dcp4) { r -= dcp4; co += 1; }
  }
}

// linear copy of n4 float4 (16-byte LDS-DMA): dst/src 16-byte aligned
__device__ __forceinline__ void glds_copy16(float* __restrict__ dst, const float* __restrict__ src, int tid, int n4) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int base = wave * 64; base < n4; base += 256) {
    if (base + lane < n4)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long)(base + lane) * 4),
                                       (__attribute__((address_space(3))) void*)(dst + (long)base * 4), 16, 0, 0);
  }
}

// The one quad per row that straddles the right limit `xend` of the readable columns is skipped by the 16-byte stager;
// edge_fix_load fetches its in-range words with ordinary loads (issued next to the DMA, so they share its latency) and
// edge_fix_store writes the whole quad (in-range words + zeros) as one 16-byte LDS store after s_waitcnt vmcnt(0).
// Image layout as in the stagers: row (ch, iy) starts at ch*chp + iy*lw; global row at (c0+ch)*H*Wg + gy*Wg.
struct EdgeFix {
  float4 v[EDGE_MAXF];
  int off[EDGE_MAXF];
};
__device__ __forceinline__ void edge_fix_load(EdgeFix& f, const float* __restrict__ src, int tid, int nch, int nrows, int lw,
                                              int chp, int c0, int y0, int x0, int C, int H, int Wg, int xend) {
  const int span = xend - x0;
  const int nvalid = span & 3;
  const bool any = span > 0 && span < lw && nvalid != 0;
  const int items = any ? nch * nrows : 0;     // one straddling quad per staged row
  const int lcol = span & ~3;                  // its first column inside the window
  const int cmax = min(nch, C - c0);
#pragma unroll
  for (int j = 0; j < EDGE_MAXF; ++j) {
    const int e = tid + 256 * j;
    f.off[j] = -1;
    f.v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < items) {
      const int ch = e / nrows, iy = e - ch * nrows;
      const int gy = y0 + iy;
      if (ch < cmax && (unsigned)gy < (unsigned)H) {     // same row test as the stager's `rowok`
        f.off[j] = ch * chp + iy * lw + lcol;
        const float* g = src + (long)(c0 + ch) * H * Wg + (long)gy * Wg + x0 + lcol;
        f.v[j].x = g[0];
        if (nvalid > 1) f.v[j].y = g[1];
        if (nvalid > 2) f.v[j].z = g[2];
      }
    }
  }
}
__device__ __forceinline__ void edge_fix_store(const EdgeFix& f, float* __restrict__ dst) {
#pragma unroll
  for (int j = 0; j < EDGE_MAXF; ++j)
    if (f.off[j] >= 0) *reinterpret_cast<float4*>(dst + f.off[j]) = f.v[j];
}

}  // namespace
