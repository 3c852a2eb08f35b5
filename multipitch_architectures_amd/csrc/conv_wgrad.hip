// Backward-weight of the convolutions other than the 15x15 ones (conv_wgrad15.hip): dW[cout][n] = sum over pixels of
// dY[cout][pixel] * X[pixel][n], n = (cin, dy, dx) flattened; A = dY tile, B = shifted input tile, workgroups loop over
// (image, tile) pairs and keep dW slices in registers; partial slices are summed by reduce_partials_kernel in a fixed
// order (no atomics).  Replaces the weight gradient of nn.Conv2d at the call sites listed in conv_fwd.hip.
#include "mpa_common.h"
#define MPA_COMMON_CDIV 1
#include "conv_plan.h"
#include "conv_stage.h"
#include "conv_internal.h"

namespace {

struct WgParams {
  const float* x;
  const float* dy;
  float* ws;
  int B, Cin, H, W, Cout, OH, OW, kh, kw, sh, sw, ph, pw;
  int COT, nPerBlock, Ntot, XCH, TH, TW, DP, tilesY, tilesX, IH, IW, LW, XCHP, DCP, S;
  int TX64, TD64;   // LDS words of the X / dY images, each rounded up to a multiple of 64
  int with_bias;   // workspace rows carry one extra column: sum over pixels of dY (the bias gradient)
  int quad, xshift;
};

template <int NBC, int NTW, bool EF = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_x = lds;
  float* lds_dy = lds + p.TX64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, l16 = lane & 15;
  const int split = blockIdx.x, ntile = blockIdx.y, cot = blockIdx.z;
  const int khkw = p.kh * p.kw;
  const int nblk0 = ntile * p.nPerBlock;
  const int ci_first = nblk0 / khkw;
  const int n_base = nblk0 + wave * NTW * 16;
  int xoff[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    int n = n_base + t * 16 + l16;
    if (n >= p.Ntot) n = nblk0;
    const int ci = n / khkw, r = n - ci * khkw;
    const int dy = r / p.kw, dx = r - dy * p.kw;
    xoff[t] = (ci - ci_first) * p.XCHP + dy * p.LW + dx + kq * p.sw + p.xshift;
  }
  const int aoff = l16 * p.DCP + kq;
  const int NtotP = p.Ntot + (p.with_bias ? 1 : 0);
  const bool do_bias = p.with_bias && ntile == 0;
  float bsum = 0.f;      // threads 2*co, 2*co+1 accumulate the bias gradient of cout co
  f32x4 acc[NBC][NTW];
#pragma unroll
  for (int i = 0; i < NBC; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tilesPerImg = p.tilesY * p.tilesX;
  const long totalTiles = (long)p.B * tilesPerImg;
  for (long tile = split; tile < totalTiles; tile += p.S) {
    const int b = (int)(tile / tilesPerImg);
    const int tr = (int)(tile - (long)b * tilesPerImg);
    const int ty = tr / p.tilesX, tx = tr - ty * p.tilesX;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;
    const int iy0 = oy0 * p.sh - p.ph, ix0 = ox0 * p.sw - p.pw;
    __syncthreads();
    // dY tile: columns >= TW belong to the neighbouring tile -> clip the readable width at ox0+TW
    if (p.quad) {
      const float* xb = p.x + (long)b * p.Cin * p.H * p.W;
      const float* db_ = p.dy + (long)b * p.Cout * p.OH * p.OW;
      const int xlim = min(p.OW, ox0 + p.TW);
      glds_stage_x16<EF>(lds_x, xb, lane, wave, p.XCH, p.IH, p.LW, p.XCHP, p.TX64, ci_first, iy0, ix0 - p.xshift, p.Cin, p.H,
                         p.W);
      glds_stage_dy16<EF>(lds_dy, db_, lane, wave, p.COT, p.TH, p.DP, p.DCP, p.TD64, cot * p.COT, oy0, ox0, p.Cout, p.OH,
                          p.OW, xlim);
      if constexpr (EF) {
        EdgeFix fx, fd;
        edge_fix_load(fx, xb, tid, p.XCH, p.IH, p.LW, p.XCHP, ci_first, iy0, ix0 - p.xshift, p.Cin, p.H, p.W, p.W);
        edge_fix_load(fd, db_, tid, p.COT, p.TH, p.DP, p.DCP, cot * p.COT, oy0, ox0, p.Cout, p.OH, p.OW, xlim);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        edge_fix_store(fx, lds_x);
        edge_fix_store(fd, lds_dy);
      }
    } else {
      glds_stage_x(lds_x, p.x + (long)b * p.Cin * p.H * p.W, lane, wave, p.XCH, p.IH, p.IW, p.LW, p.XCHP, p.TX64, ci_first,
                   iy0, ix0, p.Cin, p.H, p.W);
      glds_stage_dy(lds_dy, p.dy + (long)b * p.Cout * p.OH * p.OW, lane, wave, p.COT, p.TH, p.DP, p.DCP, p.TD64, cot * p.COT,
                    oy0, ox0, p.Cout, p.OH, p.OW, min(p.OW, ox0 + p.TW));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (do_bias) {
      // 2 threads per cout (COT <= 80 < 128), each summing half of the tile's pixels (zero-filled outside the image)
      const int co = tid >> 1, part = tid & 1;
      if (co < p.COT) {
        const float* row = lds_dy + co * p.DCP;
        const int npx = p.TH * p.DP;
        float s = 0.f;
        for (int i = part; i < npx; i += 2) s += row[i];
        bsum += s;
      }
    }
    for (int py = 0; py < p.TH; ++py) {
      const float* ap = lds_dy + py * p.DP + aoff;
      const float* bp = lds_x + py * p.sh * p.LW;
      int px0 = 0;
      if (p.sw == 1) {
        // four k-steps per trip with compile-time pixel offsets: one address VGPR per operand row and immediates for
        // the 16 pixels instead of a pointer increment per read
        for (; px0 + 16 <= p.DP; px0 += 16) {
          const float* apx[NBC];
          const float* bpx[NTW];
#pragma unroll
          for (int cb = 0; cb < NBC; ++cb) apx[cb] = ap + cb * 16 * p.DCP + px0;
#pragma unroll
          for (int t = 0; t < NTW; ++t) bpx[t] = bp + xoff[t] + px0;
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            float a[NBC], bv[NTW];
#pragma unroll
            for (int cb = 0; cb < NBC; ++cb) a[cb] = apx[cb][4 * u];
#pragma unroll
            for (int t = 0; t < NTW; ++t) bv[t] = bpx[t][4 * u];
#pragma unroll
            for (int cb = 0; cb < NBC; ++cb)
#pragma unroll
              for (int t = 0; t < NTW; ++t)
                acc[cb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb], bv[t], acc[cb][t], 0, 0, 0);
          }
        }
      }
#pragma unroll 2
      for (; px0 < p.DP; px0 += 4) {
        float a[NBC], bv[NTW];
#pragma unroll
        for (int cb = 0; cb < NBC; ++cb) a[cb] = ap[cb * 16 * p.DCP + px0];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bv[t] = bp[xoff[t] + px0 * p.sw];
#pragma unroll
        for (int cb = 0; cb < NBC; ++cb)
#pragma unroll
          for (int t = 0; t < NTW; ++t)
            acc[cb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb], bv[t], acc[cb][t], 0, 0, 0);
      }
    }
  }
  // partial slice -> workspace [split][Cout][Ntot]
  float* out = p.ws + (long)split * p.Cout * NtotP;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int n = n_base + t * 16 + l16;
    if (n >= p.Ntot) continue;
#pragma unroll
    for (int cb = 0; cb < NBC; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * p.COT + cb * 16 + kq * 4 + r;
        if (co < p.Cout) out[(long)co * NtotP + n] = acc[cb][t][r];
      }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1, 64);
    const int co = cot * p.COT + (tid >> 1);
    if ((tid & 1) == 0 && (tid >> 1) < p.COT && co < p.Cout) out[(long)co * NtotP + p.Ntot] = bsum;
  }
}


// dY-from-global variant of conv_wgrad_kernel (see conv_wgrad15g_kernel for the measurements behind it): quad geometry
// with exact tiling in x (OW % 4 == 0, TW * tilesX == OW, TW >= 16, W % 4 == 0).  A lane's float4 of dY (4 consecutive
// pixels of its cout row, one buffer load) is the A operand of 4 consecutive k-steps -- k-step j of a 16-pixel group
// contracts pixels {16g + 4kq + j} -- so the B operand of tap block t sits at xoff[t] (lane part 4kq*SW) + (16g + j)*SW:
// one address VGPR per tap block and group, immediates for j.  LDS holds the X tile only (larger pixel tiles, fewer
// halo bytes per MFMA).  A row's last DP % 16 pixels are a tail of three ordinary k-steps (pixels {4s + kq}); steps
// past the tile read zeros for A (and whatever finite words follow for B: 64 zeroed words of slack end the X region).

// NT: tail k-steps compiled in (0, 3, or 4 for rows with 13..15 pixels past the last full group).  EF: the width is not
// a multiple of 4 -- one tile per row, X staged with the row-end edge fix, dY quads only 4-byte aligned, and the lanes of
// the last tail step that lie past the row end are masked.
template <int NBC, int NTW, int NT, int SW, bool EF = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_g_kernel(const WgParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_x = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, l16 = lane & 15;
  const int split = blockIdx.x, ntile = blockIdx.y, cot = blockIdx.z;
  const int khkw = p.kh * p.kw;
  const int nblk0 = ntile * p.nPerBlock;
  const int ci_first = nblk0 / khkw;
  const int n_base = nblk0 + wave * NTW * 16;
  int xoff[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    int n = n_base + t * 16 + l16;
    if (n >= p.Ntot) n = nblk0;
    const int ci = n / khkw, r = n - ci * khkw;
    const int dy = r / p.kw, dx = r - dy * p.kw;
    xoff[t] = (ci - ci_first) * p.XCHP + dy * p.LW + dx + 4 * kq * SW + p.xshift;
  }
  const int NtotP = p.Ntot + (p.with_bias ? 1 : 0);
  const bool do_bias = p.with_bias && ntile == 0 && wave == 0;
  float bs[NBC];
  f32x4 acc[NBC][NTW];
#pragma unroll
  for (int i = 0; i < NBC; ++i) {
    bs[i] = 0.f;
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (tid < WGG_SLACK) lds_x[p.TX64 + tid] = 0.f;

  const int tilesPerImg = p.tilesY * p.tilesX;
  const long totalTiles = (long)p.B * tilesPerImg;
  const int nfull = p.TW >> 4, rem = p.TW - 16 * nfull;     // nfull >= 1; (rem + 3) / 4 <= NT tail steps
  const int plane = p.OH * p.OW;
  const int loff = (l16 * plane + 4 * kq) * 4, loff_t = (l16 * plane + kq) * 4;     // bytes
  for (long tile = split; tile < totalTiles; tile += p.S) {
    const int b = (int)(tile / tilesPerImg);
    const int tr = (int)(tile - (long)b * tilesPerImg);
    const int ty = tr / p.tilesX, tx = tr - ty * p.tilesX;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;
    const int iy0 = oy0 * p.sh - p.ph, ix0 = ox0 * p.sw - p.pw;
    const float* imgb = p.dy + (long)b * p.Cout * plane;
    const int img_elems = p.Cout * plane;
    auto dy_rsrc = [&](int cb, int py, int col, bool on) {
      const int u = (cot * p.COT + cb * 16) * plane + (oy0 + py) * p.OW + ox0 + col;
      const int left = (on && oy0 + py < p.OH && u < img_elems) ? (img_elems - u) * 4 : 0;
      return __builtin_amdgcn_make_buffer_rsrc((void*)(imgb + u), 0, left, 0x00020000);
    };
    auto load_full = [&](float4* a, int py, int g) {
#pragma unroll
      for (int cb = 0; cb < NBC; ++cb)
        a[cb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(dy_rsrc(cb, py, 16 * g, true), loff, 0, 0));
    };
    __syncthreads();
    const float* xb = p.x + (long)b * p.Cin * p.H * p.W;
    glds_stage_x16<EF>(lds_x, xb, lane, wave, p.XCH, p.IH, p.LW, p.XCHP, p.TX64, ci_first, iy0, ix0 - p.xshift, p.Cin, p.H,
                       p.W);
    EdgeFix fx;
    if constexpr (EF) edge_fix_load(fx, xb, tid, p.XCH, p.IH, p.LW, p.XCHP, ci_first, iy0, ix0 - p.xshift, p.Cin, p.H, p.W, p.W);
    float4 an[WGG_DEPTH][NBC];
#pragma unroll
    for (int d = 0; d < WGG_DEPTH; ++d) load_full(an[d], d / nfull, d % nfull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (EF) edge_fix_store(fx, lds_x);
    __syncthreads();

    for (int py = 0; py < p.TH; ++py) {
      const float* rowp = lds_x + py * p.sh * p.LW;
      float4 at[NBC];
      for (int g = 0; g < nfull; ++g) {
        float4 ac[NBC];
#pragma unroll
        for (int cb = 0; cb < NBC; ++cb) {
          ac[cb] = an[0][cb];
#pragma unroll
          for (int d = 0; d + 1 < WGG_DEPTH; ++d) an[d][cb] = an[d + 1][cb];
        }
        {
          int gd = g + WGG_DEPTH, pyd = py;
          while (gd >= nfull) { gd -= nfull; ++pyd; }
          if (pyd < p.TH) load_full(an[WGG_DEPTH - 1], pyd, gd);
        }
        if constexpr (NT > 0) {
          if (g == nfull - 1) {     // the tail's dY words: one group ahead
#pragma unroll
            for (int cb = 0; cb < NBC; ++cb) {
              at[cb].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc(cb, py, 16 * nfull, rem > 0), loff_t, 0, 0));
              at[cb].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc(cb, py, 16 * nfull + 4, rem > 4), loff_t, 0, 0));
              at[cb].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc(cb, py, 16 * nfull + 8, rem > 8), loff_t, 0, 0));
              at[cb].w = NT > 3 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc(cb, py, 16 * nfull + 12, rem > 12), loff_t, 0, 0)) : 0.f;
              if constexpr (EF) {     // pixels 16 nfull + 4s + kq >= TW belong to the next row
                if (kq >= rem) at[cb].x = 0.f;
                if (kq + 4 >= rem) at[cb].y = 0.f;
                if (kq + 8 >= rem) at[cb].z = 0.f;
                if (kq + 12 >= rem) at[cb].w = 0.f;
              }
            }
          }
        }
        if (do_bias) {
#pragma unroll
          for (int cb = 0; cb < NBC; ++cb) bs[cb] += (ac[cb].x + ac[cb].y) + (ac[cb].z + ac[cb].w);
        }
        const float* bpx[NTW];
#pragma unroll
        for (int t = 0; t < NTW; ++t) bpx[t] = rowp + xoff[t] + 16 * g * SW;
#define WGG_STEP(AV, AC, J)                                                                          \
  {                                                                                                  \
    float bv[NTW];                                                                                   \
    _Pragma("unroll") for (int t = 0; t < NTW; ++t) bv[t] = bpx[t][(J) * SW];                        \
    _Pragma("unroll") for (int cb = 0; cb < NBC; ++cb)                                               \
      _Pragma("unroll") for (int t = 0; t < NTW; ++t)                                                \
        acc[cb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[cb].AC, bv[t], acc[cb][t], 0, 0, 0);    \
    __builtin_amdgcn_sched_barrier(0);   /* keeps the scheduler from hoisting every step's reads (spills) */ \
  }
        WGG_STEP(ac, x, 0)
        WGG_STEP(ac, y, 1)
        WGG_STEP(ac, z, 2)
        WGG_STEP(ac, w, 3)
      }
      if constexpr (NT > 0) {
        if (do_bias) {
#pragma unroll
          for (int cb = 0; cb < NBC; ++cb) bs[cb] += (at[cb].x + at[cb].y) + (at[cb].z + at[cb].w);
        }
        const float* bpx[NTW];       // lane part kq*SW instead of 4kq*SW
#pragma unroll
        for (int t = 0; t < NTW; ++t) bpx[t] = rowp + xoff[t] + (16 * nfull - 3 * kq) * SW;
        WGG_STEP(at, x, 0)
        WGG_STEP(at, y, 4)
        WGG_STEP(at, z, 8)
        if constexpr (NT > 3) WGG_STEP(at, w, 12)
      }
#undef WGG_STEP
    }
  }
  // partial slice -> workspace [split][Cout][Ntot]
  float* out = p.ws + (long)split * p.Cout * NtotP;
#pragma unroll
  for (int t = 0; t < NTW; ++t) {
    const int n = n_base + t * 16 + l16;
    if (n >= p.Ntot) continue;
#pragma unroll
    for (int cb = 0; cb < NBC; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * p.COT + cb * 16 + kq * 4 + r;
        if (co < p.Cout) out[(long)co * NtotP + n] = acc[cb][t][r];
      }
  }
  if (do_bias) {
#pragma unroll
    for (int cb = 0; cb < NBC; ++cb) {
      float v = bs[cb];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int co = cot * p.COT + cb * 16 + l16;
      if (kq == 0 && co < p.Cout) out[(long)co * NtotP + p.Ntot] = v;
    }
  }
}


// ws [S][Cout][NtotP] -> dw [Cout][Ntot] (+ db [Cout] from the extra column)
__global__ void reduce_partials_kernel(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db,
                                       int Cout, int Ntot, int NtotP, int S) {
  const long n = (long)Cout * NtotP;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // eight independent chains: eight loads in flight per thread (one chain ran at a sixth of the HBM rate); the order
    // of the additions is fixed, so the result stays run-to-run reproducible
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.f;
    const float* src = ws + i;
    int k = 0;
    for (; k + 8 <= S; k += 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] += src[(long)(k + u) * n];
    }
    for (; k < S; ++k) a[k & 7] += src[(long)k * n];
    const float s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const int co = (int)(i / NtotP), j = (int)(i - (long)co * NtotP);
    if (j < Ntot) dw[(long)co * Ntot + j] = s;
    else if (db) db[co] = s;
  }
}

}  // namespace

int mpa_conv_reduce_partials(const float* ws, float* dw, float* db, int Cout, int Ntot, int NtotP, int S, hipStream_t s) {
  const long n = (long)Cout * NtotP;
  MPA_LAUNCH(reduce_partials_kernel, dim3((unsigned)std::min<long>(mpa_cdiv(n, 256), 2048)), dim3(256), 0, s, ws, dw, db, Cout,
             Ntot, NtotP, S);
  return mpa_launch_status();
}

int mpa_conv_wgrad_generic(const mpa_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* workspace,
                           int64_t workspace_bytes, hipStream_t s) {
  WgPlan pl = plan_wgrad(d);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  const int64_t need = (int64_t)pl.S * d->Cout * (pl.Ntot + 1) * 4;
  if (!workspace || workspace_bytes < need) return MPA_ERR_WORKSPACE;
  WgParams p{};
  p.x = x; p.dy = dy; p.ws = (float*)workspace;
  p.B = d->B; p.Cin = d->Cin; p.H = d->H; p.W = d->W; p.Cout = d->Cout; p.OH = pl.OH; p.OW = pl.OW;
  p.kh = d->kh; p.kw = d->kw; p.sh = d->sh; p.sw = d->sw; p.ph = d->ph; p.pw = d->pw;
  p.COT = pl.COT; p.nPerBlock = pl.nPerBlock; p.Ntot = pl.Ntot; p.XCH = pl.XCH; p.TH = pl.TH; p.TW = pl.TW; p.DP = pl.DP;
  p.tilesY = pl.tilesY; p.tilesX = pl.tilesX; p.IH = pl.IH; p.IW = pl.IW; p.LW = pl.LW; p.XCHP = pl.XCHP; p.DCP = pl.DCP;
  p.S = pl.S;
  p.TX64 = (int)(mpa_cdiv((long)pl.XCH * pl.XCHP, 64) * 64);
  p.TD64 = (int)(mpa_cdiv((long)pl.COT * pl.DCP, 64) * 64);
  p.with_bias = 1;
  p.quad = pl.quad; p.xshift = pl.xshift;
  dim3 grid((unsigned)pl.S, (unsigned)pl.nTiles, (unsigned)pl.coTiles);
#define MPA_WG_LAUNCH(NBC_, NTW_)                                                                            \
  do {                                                                                                      \
    if (pl.ef) MPA_LAUNCH((conv_wgrad_kernel<NBC_, NTW_, true>), grid, dim3(256), pl.lds_bytes, s, p);      \
    else MPA_LAUNCH((conv_wgrad_kernel<NBC_, NTW_, false>), grid, dim3(256), pl.lds_bytes, s, p);           \
  } while (0)
#define MPA_WGG_LAUNCH(NBC_, NTW_, SW_)                                                                      \
  do {                                                                                                      \
    if (pl.TW & 15) MPA_LAUNCH((conv_wgrad_g_kernel<NBC_, NTW_, 3, SW_>), grid, dim3(256), pl.lds_bytes, s, p);      \
    else MPA_LAUNCH((conv_wgrad_g_kernel<NBC_, NTW_, 0, SW_>), grid, dim3(256), pl.lds_bytes, s, p);        \
  } while (0)
  if (pl.ga && pl.ef) {        // unaligned rows: <2,8>, stride 1
    const int rem = pl.TW & 15;
    if (rem > 12) MPA_LAUNCH((conv_wgrad_g_kernel<2, 8, 4, 1, true>), grid, dim3(256), pl.lds_bytes, s, p);
    else if (rem) MPA_LAUNCH((conv_wgrad_g_kernel<2, 8, 3, 1, true>), grid, dim3(256), pl.lds_bytes, s, p);
    else MPA_LAUNCH((conv_wgrad_g_kernel<2, 8, 0, 1, true>), grid, dim3(256), pl.lds_bytes, s, p);
  } else
  if (pl.ga) {
    if (d->sw == 3) MPA_WGG_LAUNCH(5, 6, 3);
    else if (pl.NBC == 1) MPA_WGG_LAUNCH(1, 16, 1);
    else if (pl.NBC == 2 && pl.NTW == 16) MPA_WGG_LAUNCH(2, 16, 1);
    else if (pl.NBC == 2) MPA_WGG_LAUNCH(2, 8, 1);
    else if (pl.NBC == 4) MPA_WGG_LAUNCH(4, 6, 1);
    else MPA_WGG_LAUNCH(5, 6, 1);
  } else
  if (pl.NBC == 1) MPA_WG_LAUNCH(1, 16);
  else if (pl.NBC == 2 && pl.NTW == 16) MPA_WG_LAUNCH(2, 16);
  else if (pl.NBC == 2) MPA_WG_LAUNCH(2, 8);
  else if (pl.NBC == 4) MPA_WG_LAUNCH(4, 6);
  else MPA_WG_LAUNCH(5, 6);
#undef MPA_WG_LAUNCH
#undef MPA_WGG_LAUNCH
  int rc = mpa_launch_status();
  if (rc) return rc;
  return mpa_conv_reduce_partials((const float*)workspace, dw, db, d->Cout, pl.Ntot, pl.Ntot + 1, pl.S, s);
}
