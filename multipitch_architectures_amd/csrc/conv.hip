// Implicit-GEMM 2-D convolution for gfx950 on v_mfma_f32_16x16x4_f32 (exact fp32).
//
// Replaces nn.Conv2d at: double_conv (unet_cnns.py:49-59), conv1/prefilt_list
// (basic_cnns.py:371-387), conv2/conv3/conv4 (unet_cnns.py:538-557), convP (:2311-2318).
//
// Forward   D[cout][pixel] = sum_k Wp[cout][k] * X[k][pixel],  k = (cin, dy, dx)
//   MFMA A operand = filters (m = cout), B operand = input halo tile in LDS
//   (n = pixel), so the accumulator's lane index runs along pixels -> coalesced
//   NCHW stores.  One block = 4 waves; a wave owns NB cout-blocks x PB
//   pixel-blocks of 16x16.  The input tile (CK channels + halo) is staged once
//   per channel chunk, the filter slab (one dy row: kw*CK*COT floats) once per
//   (chunk, dy).
// Backward-data = forward with flipped / transposed filters (pack mode 1).
// Backward-weight: dW[cout][n] = sum_pixel dY[cout][pixel] * X[pixel][n],
//   n = (cin,dy,dx) flattened; A = dY tile, B = shifted input tile, blocks loop
//   over (image, tile) pairs and keep dW slices in registers; partials are
//   reduced by a second kernel (deterministic, no atomics).
#include "mpa_common.h"
#include <algorithm>
#include <cstring>
#include <cmath>
#include <cstdio>
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------------ planning
constexpr long FWD_LDS_BUDGET = 78 * 1024;   // two workgroups per CU (160 KiB LDS)
constexpr int EDGE_MAXF = 3;                 // straddling quads (rows) per thread that edge_fix_* can carry (see there)

struct FwdPlan {
  int NB, PB, TH, TW, tilesY, tilesX, CK, nChunks, IH, IW, LW, CHP, COT, COTP, coTiles, OH, OW, quad;
  size_t lds_bytes;
  bool ok;
  int KWS, KWP;   // tap-vector filter layout (kw 15 / 9, PB >= 4): kernel specialised on kw, slab rows of KWP taps
  int KS;         // input-channel split: blockIdx.z owns nChunks/KS chunks and adds its partial sums atomically
};

// Which problems use the tap-vector layout [ck][cout][dx padded to KWP]: the A operand of 4 consecutive taps is then one
// ds_read_b128 and every B read is base + immediate, i.e. ~0.6 instead of ~1.2 non-MFMA vector instructions per MFMA for
// the 16-cout kernels (each such instruction costs the SIMD about 4 of the 32 cycles an MFMA occupies).
// Measured: +4..8 % for NB <= 2; the 64-cout tile (NB = 4, already at 1 operand read per 2.7 MFMAs) loses 5 % to the
// extra live registers, so it keeps the tap-major layout.
inline int fwd_kw_special(int kw, int NB, int PB) {
  return ((kw == 15 || kw == 9 || kw == 5 || kw == 3) && PB >= 4 && (NB <= 2 || NB * PB <= 30)) ? kw : 0;   // <4,8> would need > 256 VGPRs
}

inline int round_mod(int v, int m, int r) {  // smallest x >= v with x % m == r
  int x = v + ((r - v % m) % m + m) % m;
  return x;
}

// allow_split: the launch may add channel slices atomically (backward-data only: the forward pass stays bit-reproducible)
// phase: the launch stores through the (channel, phase) mapping -- only built for the generic and 15-tap loops, without
// the row-end edge fix
// phaseX == 3 (stride-(1,3) backward-data): 48-cout tiles = 16 channels x 3 phases, so that a workgroup owns whole
// (channel, pixel) triples and can store them as contiguous 16-byte runs
FwdPlan plan_fwd(int B, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                 bool allow_split = false, bool phase = false, int phaseX = 1) {
  FwdPlan best{};
  best.ok = false;
  const int OH = (H + 2 * ph - kh) / sh + 1, OW = (W + 2 * pw - kw) / sw + 1;
  if (OH <= 0 || OW <= 0) return best;
  const int cin4 = (int)mpa_cdiv(Cin, 4) * 4;
  double bestcost = 1e300;
  // cout blocking: all variants compete; the cost model charges padded couts, operand re-reads, and -- what decides
  // small batches / small images -- the number of *rounds* the grid needs on 256 CUs (a grid of 2112 workgroups on 512
  // resident slots costs 5 rounds, not 4.1)
  const int nbs[6] = {1, 2, 4, 5, 3, 6};
  const int pbs[6] = {1, 2, 4, 6, 8, 12};
  static const char* force = getenv("MPA_FWD_FORCE");          // diagnostics: "NB,PB" restricts the search
  int fNB = 0, fPB = 0;
  if (force) sscanf(force, "%d,%d", &fNB, &fPB);
  for (int ni = 0; ni < 6; ++ni) {
    const int NB = nbs[ni];
    if (fNB && NB != fNB) continue;
    // 48- / 96-cout tiles (16 / 32 channels x 3 phases) exactly for the 3-phase stores
    if ((NB % 3 == 0) != (phase && phaseX == 3)) continue;
    const int COT = NB * 16;
    const int coTiles = (int)mpa_cdiv(Cout, COT);
    if (ni > 0 && (long)coTiles * COT > (long)mpa_cdiv(Cout, 16) * 16 + 32 && NB > 1) continue;   // too much cout padding
    const int COTP = (COT % 32 == 0) ? COT + 16 : COT;   // filter-slab pitch == 16 (mod 32): conflict-free A reads
    for (int pi = 0; pi < 6; ++pi) {
      if ((pbs[pi] == 12 && NB > 2) || (pbs[pi] == 8 && NB > 4)) continue;   // accumulator budget
      if (NB == 3 && (pbs[pi] < 4 || pbs[pi] > 8)) continue;                  // built for PB 4, 6, 8 only
      if (NB == 6 && pbs[pi] != 4) continue;                                  // 24 accumulator tiles
      const int PB = pbs[pi], P = PB * 64;
      if (fPB && PB != fPB) continue;
      for (int TH = 1; TH <= std::min(OH, P); ++TH) {
        const int TWmax = std::min(OW, P / TH);
        if (TWmax < 1) continue;
        const int tx0 = (int)mpa_cdiv(OW, TWmax);
        for (int txi = 0; txi < 10; ++txi) {
          // candidate tile widths: the widest that fits, then progressively narrower ones (LDS-limited tall kernels)
          const int tx = txi < 6 ? tx0 + txi : tx0 << (txi - 4);
          if (tx > OW) break;
          const int TW = (int)mpa_cdiv(OW, tx);
          const int ty = (int)mpa_cdiv(OH, TH);
          const int IH = (TH - 1) * sh + kh, IW = (TW - 1) * sw + kw;
          for (int lwi = 0; lwi < 2; ++lwi) {
            // row pitch == TW (mod 32) keeps pixel blocks that wrap a row conflict-free; fall back to the tight pitch.
            // 16-byte LDS-DMA staging (quad): stride 1, pitch % 4 == 0 and 3 spare columns for the 4-aligned window
            // origin; when W % 4 != 0 the quad straddling the end of each row is completed by edge_fix_* (bounded
            // number of such words per workgroup).
            int LW = (lwi == 0 && sw == 1 && kw - 1 <= 29) ? TW + 32 : (IW | 1);
            if (LW < IW) LW = IW | 1;
            int quad = 0;
            {                      // (the window is a contiguous block of columns whatever the stride of the taps)
              int lq = LW;
              if (lq % 4 != 0 || lq < IW + 3) lq = (int)mpa_cdiv(std::max(LW, IW + 3), 4) * 4;
              if (lwi == 1 || lq == LW) { LW = lq; quad = 1; }
            }
            const int CHP = round_mod(IH * LW, 32, 16);
            int CK = 4;
            while (CK < 32 && CK < cin4 && kw * (CK / 4) < 15) CK *= 2;
            int KWS = fwd_kw_special(kw, NB, PB);
            if ((phase && KWS != 15) || NB % 3 == 0) KWS = 0;
            const int KWP = (kw + 3) & ~3;
            const int cotp = KWS ? COT : COTP;
            auto lds_words = [&](int ck) {
              const long slab = KWS ? (long)ck * COT * KWP : (long)kw * ck * COTP;
              return mpa_cdiv((long)ck * CHP, 64) * 64 + 2 * (mpa_cdiv(slab, 64) * 64);
            };
            while (CK > 4 && lds_words(CK) * 4 > 52 * 1024) CK /= 2;   // keep three workgroups per CU when the chunk allows
            const size_t lds = (size_t)lds_words(CK) * 4;
            if ((long)lds > FWD_LDS_BUDGET) continue;
            if (quad && (W & 3) && ((long)CK * IH > 256 * EDGE_MAXF || KWS >= 9 || phase)) quad = 0;   // cannot fix up: dword staging
            // resident workgroups per CU: LDS and (estimated) VGPR limits
            const int regs = NB * PB * 4 + 4 * (NB + PB) + 48;
            const long bpc = std::max<long>(1, std::min<long>(std::min<long>(4, (160 * 1024) / (long)lds), 512 / regs));
            const long blocks = (long)B * ty * tx * coTiles;
            // cycles one workgroup needs when it shares each SIMD with bpc-1 others
            // operand term: every non-MFMA vector instruction costs ~4 of an MFMA's 32 cycles -- (NB+PB) LDS reads plus
            // their address arithmetic per NB*PB MFMAs (about half of that in the tap-vector kernels); barrier term:
            // ~400 cycles per (chunk, filter row) against kw*(CK/4)*NB*PB MFMAs of 32 cycles
            const double opnd = (KWS ? 0.13 : 0.25) * (NB + PB) / (double)(NB * PB);
            const double per_block = (double)P * NB * (1.0 + 0.05 * IH * IW / P + opnd) * (1.0 + 0.02 * lwi) *
                                     (1.0 + 12.5 / ((double)kw * (CK / 4) * NB * PB)) * (quad ? 1.0 : 1.08);
            // large grids: throughput (blocks * per_block / 256 CUs); small grids: whole rounds
            const double fill = (double)(TH * TW) / P;      // lanes doing useful work
            // A CU's MFMA pipes are shared by its resident workgroups, so what a launch costs is the number of
            // workgroups the *busiest CU* has to work through: ceil(blocks / 256) tiles for small grids (704 tiles are
            // 2.75 per CU, i.e. 3 -- measured 117 instead of 132 TFLOP/s for the 128->16 backward-data at batch 32,
            // where 1024 smaller tiles are exactly 4 per CU), blocks / 256 plus a drifting tail for large ones.
            const double xcu = (double)blocks / 256.0;
            auto cu_load = [&](double x) { return x <= 3.0 * bpc ? std::ceil(x) : x + 0.35 * bpc; };
            double cost = cu_load(xcu) * per_block * (1.0 + 0.25 * (1.0 - fill)) + 1e-3 * blocks;
            // Small grids (small batch x small image: the U-Net's deep levels): split the input channels over
            // blockIdx.z so that efficient wave tiles still fill the chip; partial sums are added atomically into a
            // zeroed output.  Each extra slice pays a prologue/epilogue (~6 % of a full-K workgroup).
            const int nChunks = (int)mpa_cdiv(Cin, CK);
            int KS = 1;
            static const int ks_max = getenv("MPA_FWD_KS_MAX") ? atoi(getenv("MPA_FWD_KS_MAX")) : 16;   // diagnostics
            static const int ks_force = getenv("MPA_FWD_KS_FORCE") ? atoi(getenv("MPA_FWD_KS_FORCE")) : 0;   // diagnostics
            if (ks_force > 1 && allow_split && ks_force <= nChunks) KS = ks_force;
            else
            if (xcu <= 2.0 * bpc && allow_split) {
              // Each extra slice pays a prologue / epilogue: ~6 % of a full-K workgroup for the short reductions of the
              // deep levels, next to nothing for 32 chunks of 15x15 taps.  Slices of a grid that is a bad fraction of
              // the chip level it out: 704 tiles are 2.75 per CU (the busiest CU works through 3), 4 x 704 quarter
              // tiles are exactly 11 per CU -- measured 3.97 -> 3.61 ms for the 128->16 backward-data at local batch
              // 32 (scratch/ks_force.sh), i.e. whole quarter tiles and no drifting tail up to ~6 resident sets.
              const double slice_cost = std::min(0.06, 60.0 / ((double)nChunks * kh * kw * (CK / 4)));
              for (int ks = 2; ks <= ks_max && ks <= nChunks; ks *= 2) {
                const double load = xcu * ks <= 6.0 * bpc ? std::ceil(xcu * ks) : cu_load(xcu * ks);
                const double c2 = load * per_block * (1.0 / ks + slice_cost) * (1.0 + 0.25 * (1.0 - fill)) + 1e-3 * blocks * ks;
                if (c2 < cost) { cost = c2; KS = ks; }
              }
            }
            if (cost < bestcost) {
              bestcost = cost;
              best = FwdPlan{NB, PB, TH, TW, ty, tx, CK, nChunks, IH, IW, LW, CHP, COT, cotp, coTiles, OH, OW,
                             quad, lds, true, KWS, KWP, KS};
            }
          }
        }
      }
    }
  }
  if (!best.ok && phaseX == 3)                            // no 48-cout tiling fits: ordinary tiles, scalar phase stores
    return plan_fwd(B, Cin, H, W, Cout, kh, kw, sh, sw, ph, pw, allow_split, phase, 1);
  return best;
}

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

struct ConvFwdParams {
  const float* x;
  const float* wp;
  const float* bias;
  float* y;
  int B, Cin, H, W, Cout, OH, OW, kh, kw, sh, sw, ph, pw;
  int TH, TW, tilesY, tilesX, CK, nChunks, IH, IW, LW, CHP, COT, COTP;
  int IN64, SL64;      // LDS words of the input tile / one filter slab, rounded up to multiples of 64
  int quad;            // 16-byte LDS-DMA staging of the input tile (window origin rounded down to a multiple of 4)
  int dbg;             // diagnostics (env MPA_DEBUG_FWD): 1 = stage only once, 2 = skip the MFMA loops
  int act;
  float slope;
  long outBS, outCS;   // output batch / channel strides (floats)
  int outRS, outXmul, outCdiv;
  int outYmul, outH;   // phase stores: cout' = cin*(outXmul*outYmul) + v*outXmul + q -> row oy*outYmul+v (< outH), column ox*outXmul+q
  int chunksPer;       // input-channel chunks per blockIdx.z slice (== nChunks when the channels are not split)
  int coTiles, nTilesAll;   // cout tiles; pixel tiles over the whole batch
  float* stats;             // BatchNorm fusion: per-(pixel tile, cout) partial sums of y and y^2 -> [nTilesAll][Cout][2]
};

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

// ------------------------------------------------------------------------------------------------ forward kernel
// PH (phase stores): backward-data variants whose couts are (channel, x/y phase) pairs -- a separate instantiation, as
// EF is: the 16->128 forward sits on a register cliff and lost 10 % whenever either was compiled into the common kernel
template <int NB, int PB, int KW = 0, bool EF = false, bool PH = false>
__global__ __launch_bounds__(256) void conv_fwd_kernel(const ConvFwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_in = lds;
  float* lds_w0 = lds + p.IN64;          // two filter-slab buffers: slab dy+1 streams in while dy is consumed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so workgroup w runs
  // on XCD w%8 as the (w/8)-th of that XCD.  The coTiles workgroups that read the same input tile are made consecutive
  // *within one XCD*: the tile is fetched from HBM once and hit in that L2 by the others.
  const int w = blockIdx.x;
  const int seq = w >> 3;
  const int cot = seq % p.coTiles;
  int bid = (seq / p.coTiles) * 8 + (w & 7);
  if (bid >= p.nTilesAll) return;          // padding of the last group of 8 pixel tiles (whole workgroup)
  const int ptile = bid;
  const int tx = bid % p.tilesX;
  bid /= p.tilesX;
  const int ty = bid % p.tilesY;
  const int b = bid / p.tilesY;
  const int oy0 = ty * p.TH, ox0 = tx * p.TW;
  const int iy0 = oy0 * p.sh - p.ph, ix0 = ox0 * p.sw - p.pw;
  const int npix = p.TH * p.TW;
  const int kq = lane >> 4, l16 = lane & 15;

  const int x0a = p.quad ? (ix0 & ~3) : ix0;     // 4-aligned window origin for the 16-byte staging path
  const int xshift = ix0 - x0a;
  int boff[PB];
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    int pix = (wave * PB + pb) * 16 + l16;
    int pc = pix < npix ? pix : npix - 1;
    int py = pc / p.TW, px = pc - py * p.TW;
    boff[pb] = kq * p.CHP + py * p.sh * p.LW + px * p.sw + xshift;
  }
  const int aoff = kq * p.COTP + l16;
  f32x4 acc[NB][PB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float* xb = p.x + (long)b * p.Cin * p.H * p.W;
  constexpr int KWP = (KW + 3) & ~3;
  const int slab = KW ? p.CK * p.COTP * KWP : p.kw * p.CK * p.COTP;
  // tap-vector layout: lane (kq, l16) owns the KWP-tap row of (channel kq, cout l16); with 16-tap rows the four 16-byte
  // chunks of a row are rotated by l16>>2 (done by the packer) so that 16 lanes hit 16 disjoint bank quads
  int arow[KWP / 4 > 0 ? KWP / 4 : 1];
  if constexpr (KW > 0) {
#pragma unroll
    for (int g = 0; g < KWP / 4; ++g)
      arow[g] = (kq * p.COTP + l16) * KWP + (KWP == 16 ? ((g + (l16 >> 2)) & 3) * 4 : g * 4);
  }
  const float* wtile = p.wp + (long)cot * p.nChunks * p.kh * slab;
  const int astep = p.CK * p.COTP;

  EdgeFix efix;            // only live in the EF instantiations
  (void)efix;
  const bool split = gridDim.z > 1;
  const int c_begin = blockIdx.z * p.chunksPer, c_end = min(p.nChunks, c_begin + p.chunksPer);
  for (int c = c_begin; c < c_end; ++c) {
    __syncthreads();   // every wave is done with the previous chunk's tile and slabs
    const bool do_stage = (p.dbg != 1 && p.dbg != 3) || c == c_begin;
    if (do_stage) {
      if (p.quad) {
        glds_stage_x16<EF>(lds_in, xb, lane, wave, p.CK, p.IH, p.LW, p.CHP, p.IN64, c * p.CK, iy0, x0a, p.Cin, p.H, p.W);
        if constexpr (EF) edge_fix_load(efix, xb, tid, p.CK, p.IH, p.LW, p.CHP, c * p.CK, iy0, x0a, p.Cin, p.H, p.W, p.W);
      } else
        glds_stage_x(lds_in, xb, lane, wave, p.CK, p.IH, p.IW, p.LW, p.CHP, p.IN64, c * p.CK, iy0, ix0, p.Cin, p.H, p.W);
      glds_copy16(lds_w0, wtile + (long)(c * p.kh) * slab, tid, slab / 4);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (EF) { if (p.quad && do_stage) edge_fix_store(efix, lds_in); }
    __syncthreads();
    for (int dy = 0; dy < p.kh; ++dy) {
      const float* lds_w = lds_w0 + (dy & 1) * p.SL64;
      if (dy + 1 < p.kh && do_stage)
        glds_copy16(lds_w0 + ((dy + 1) & 1) * p.SL64, wtile + (long)(c * p.kh + dy + 1) * slab, tid, slab / 4);
      if constexpr (KW > 0) {
        // kw = 15 always runs with 4-channel chunks (plan_fwd), so the channel-group loop has a single trip there
        const int nj = KW == 15 ? 1 : p.CK / 4;
        if (p.dbg != 2)
        for (int j = 0; j < nj; ++j) {
          const float* aw = lds_w + (KW == 15 ? 0 : j * 4 * p.COTP * KWP);
          const float* bp = lds_in + (KW == 15 ? 0 : j * 4 * p.CHP) + dy * p.LW;
#pragma unroll
          for (int g = 0; g < KWP / 4; ++g) {
            const int taps = KW - 4 * g >= 4 ? 4 : KW - 4 * g;
            f32x4 a4[NB];
            float bv[4][PB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a4[nb] = *(const f32x4*)(aw + arow[g] + nb * 16 * KWP);
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (u < taps) {
#pragma unroll
                for (int pb = 0; pb < PB; ++pb) bv[u][pb] = bp[boff[pb] + 4 * g + u];
              }
#pragma unroll
            for (int u = 0; u < 4; ++u)
              if (u < taps) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                  for (int pb = 0; pb < PB; ++pb)
                    acc[nb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[nb][u], bv[u][pb], acc[nb][pb], 0, 0, 0);
              }
          }
        }
      } else
      if (p.dbg != 2)
      for (int j = 0; j < p.CK / 4; ++j) {
        const float* ap = lds_w + j * 4 * p.COTP + aoff;
        const float* bp = lds_in + j * 4 * p.CHP + dy * p.LW;
        // taps in groups of 3 (kw = 15, 9, 3 for every large filter of the model): all operand reads of a group are
        // issued before its MFMAs, the remaining latency is covered by the other resident waves
        int dx = 0;
        for (; dx + 3 <= p.kw; dx += 3) {
          float a[3][NB], bv[3][PB];
#pragma unroll
          for (int u = 0; u < 3; ++u) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) a[u][nb] = ap[(dx + u) * astep + nb * 16];
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) bv[u][pb] = bp[boff[pb] + dx + u];
          }
#pragma unroll
          for (int u = 0; u < 3; ++u)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
#pragma unroll
              for (int pb = 0; pb < PB; ++pb)
                acc[nb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][nb], bv[u][pb], acc[nb][pb], 0, 0, 0);
        }
        for (; dx < p.kw; ++dx) {
          float a[NB], bv[PB];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) a[nb] = ap[dx * astep + nb * 16];
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) bv[pb] = bp[boff[pb] + dx];
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb)
              acc[nb][pb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[nb], bv[pb], acc[nb][pb], 0, 0, 0);
        }
      }
      if (dy + 1 < p.kh && p.dbg != 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next slab has landed
        __syncthreads();                                   // ... and everyone is done reading this one
      }
    }
  }

  // epilogue: lane holds 4 consecutive couts (rows) of one pixel (column).
  // Wide path (NB*PB > 16, plain NCHW target, TW % 4 == 0, OW % 4 == 0): every 16x16 accumulator tile is transposed
  // through a wave-private LDS patch so that a lane owns 4 consecutive pixels of one cout and writes one 16-byte store
  // -- 4x fewer store instructions (measured: the 128 dword stores per lane of <4,8> cost 6.4 % of the workgroup's life).
  if (!PH && !split && NB * PB >= 12 && p.outCdiv >= p.Cout && (p.TW & 3) == 0 && (p.OW & 3) == 0 && p.act != MPA_ACT_SIGMOID) {
    __syncthreads();                                  // the main loop's LDS images are dead now
    float* patch = lds + wave * (16 * 20);            // [cout 16][pixel 16 (+4 pad)]
    const int co_l = lane >> 2, quad = lane & 3;      // after the transpose: lane -> (cout row, 4-pixel group)
    const float neg_scale = p.act == MPA_ACT_NONE ? 1.f : (p.act == MPA_ACT_RELU ? 0.f : p.slope);
    float* yb = p.y + (long)b * p.outBS;
    float ssum[NB], qsum[NB];                         // BatchNorm partials (p.stats): this lane's 4-pixel groups of cout co_l
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) { ssum[nb] = 0.f; qsum[nb] = 0.f; }
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int pix4 = (wave * PB + pb) * 16 + quad * 4;
      const int pc = pix4 < npix ? pix4 : 0;
      const int py = pc / p.TW, px = pc - py * p.TW;
      const int oy = oy0 + py, ox = ox0 + px;
      const bool ok4 = pix4 < npix && oy < p.OH && ox < p.OW;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[(kq * 4 + r) * 20 + l16] = acc[nb][pb][r];
        __builtin_amdgcn_wave_barrier();
        float4 v = *reinterpret_cast<const float4*>(patch + co_l * 20 + quad * 4);
        __builtin_amdgcn_wave_barrier();
        const int co = cot * p.COT + nb * 16 + co_l;
        const float bs = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
        v.x += bs; v.y += bs; v.z += bs; v.w += bs;
        if (p.stats && ok4) {                         // (act is NONE in front of a BatchNorm)
          ssum[nb] += (v.x + v.y) + (v.z + v.w);
          qsum[nb] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        }
        v.x = v.x >= 0.f ? v.x : v.x * neg_scale; v.y = v.y >= 0.f ? v.y : v.y * neg_scale;
        v.z = v.z >= 0.f ? v.z : v.z * neg_scale; v.w = v.w >= 0.f ? v.w : v.w * neg_scale;
        if (ok4 && co < p.Cout)
          *reinterpret_cast<float4*>(yb + (long)co * p.outCS + (long)oy * p.outRS + ox) = v;
      }
    }
    if (p.stats) {
      // lane quads -> one value per (wave, cout), waves -> workgroup in a fixed order, one row of partials per pixel tile
      float* red = lds + 4 * (16 * 20);               // [wave][COT][2], behind the four transpose patches
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        float sv = ssum[nb], qv = qsum[nb];
        sv += __shfl_xor(sv, 1, 64); qv += __shfl_xor(qv, 1, 64);
        sv += __shfl_xor(sv, 2, 64); qv += __shfl_xor(qv, 2, 64);
        if (quad == 0) {
          red[(wave * p.COT + nb * 16 + co_l) * 2] = sv;
          red[(wave * p.COT + nb * 16 + co_l) * 2 + 1] = qv;
        }
      }
      __syncthreads();
      const int co = cot * p.COT + tid;
      if (tid < p.COT && co < p.Cout) {
        const float s4 = (red[tid * 2] + red[(p.COT + tid) * 2]) + (red[(2 * p.COT + tid) * 2] + red[(3 * p.COT + tid) * 2]);
        const float q4 = (red[tid * 2 + 1] + red[(p.COT + tid) * 2 + 1]) +
                         (red[(2 * p.COT + tid) * 2 + 1] + red[(3 * p.COT + tid) * 2 + 1]);
        *reinterpret_cast<float2*>(p.stats + ((long)ptile * p.Cout + co) * 2) = make_float2(s4, q4);
      }
    }
    return;
  }
  if constexpr (PH && NB % 3 == 0) {
    // stride-(1,3) backward-data with 48- or 96-cout tiles: cout' = 3*channel + phase, NB / 3 groups of 16 channels.  The
    // three tiles of a group and pixel block go through a wave-private LDS patch; a lane then owns (channel, 4 pixels)
    // = 12 consecutive floats of dx and writes them as three 16-byte stores (the scalar path scatters 4-byte stores 12
    // bytes apart).  96-cout tiles halve the number of workgroups that stage the same dY tile.
    if (!split && p.outXmul == 3 && p.outYmul == 1 && (p.TW & 3) == 0 && (p.OW & 3) == 0 && ((p.outRS * 3) & 3) == 0) {
      __syncthreads();
      float* patch = lds + wave * (48 * 20);           // [cout' 48][pixel 16 (+4 pad)]
      const int cc_l = lane >> 2, quad = lane & 3;
#pragma unroll
      for (int grp = 0; grp < NB / 3; ++grp) {
        const int cc = cot * (NB / 3) * 16 + grp * 16 + cc_l;      // channel of dx
        float* yb = p.y + (long)b * p.outBS + (long)cc * p.outCS;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
          const int pix4 = (wave * PB + pb) * 16 + quad * 4;
          const int pc = pix4 < npix ? pix4 : 0;
          const int py = pc / p.TW, px = pc - py * p.TW;
          const int oy = oy0 + py, ox = ox0 + px;
          const bool ok4 = pix4 < npix && oy < p.OH && ox < p.OW && cc < p.outCdiv;
#pragma unroll
          for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) patch[(nb * 16 + kq * 4 + r) * 20 + l16] = acc[grp * 3 + nb][pb][r];
          __builtin_amdgcn_wave_barrier();
          float o[12];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const float4 t = *reinterpret_cast<const float4*>(patch + (cc_l * 3 + q) * 20 + quad * 4);
            o[q] = t.x; o[3 + q] = t.y; o[6 + q] = t.z; o[9 + q] = t.w;
          }
          __builtin_amdgcn_wave_barrier();
          if (ok4) {
            float* dst = yb + (long)oy * p.outRS + (long)ox * 3;
            *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(o[4], o[5], o[6], o[7]);
            *reinterpret_cast<float4*>(dst + 8) = make_float4(o[8], o[9], o[10], o[11]);
          }
        }
      }
      return;
    }
  }
  if constexpr (!PH) {
    if (p.stats) {
      // BatchNorm partials from the accumulators (+ bias): lane (kq, l16) holds couts nb*16 + kq*4 + r of pixel l16 of
      // each of its PB blocks; sum its valid pixels, then the 16 pixel lanes, then the four waves through LDS
      __syncthreads();                                  // the main loop's LDS images are dead now
      float* red = lds;                                 // [wave][COT][2]
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = nb * 16 + kq * 4 + r, co = cot * p.COT + col;
          const float bs = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
          float sv = 0.f, qv = 0.f;
#pragma unroll
          for (int pb = 0; pb < PB; ++pb) {
            const int pix = (wave * PB + pb) * 16 + l16;
            const int pc = pix < npix ? pix : 0;
            const int py = pc / p.TW, px = pc - py * p.TW;
            const bool ok = pix < npix && oy0 + py < p.OH && ox0 + px < p.OW;
            const float v = acc[nb][pb][r] + bs;
            if (ok) { sv += v; qv += v * v; }
          }
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }
          if (l16 == 0) { red[(wave * p.COT + col) * 2] = sv; red[(wave * p.COT + col) * 2 + 1] = qv; }
        }
      }
      __syncthreads();
      const int co = cot * p.COT + tid;
      if (tid < p.COT && co < p.Cout) {
        const float s4 = (red[tid * 2] + red[(p.COT + tid) * 2]) + (red[(2 * p.COT + tid) * 2] + red[(3 * p.COT + tid) * 2]);
        const float q4 = (red[tid * 2 + 1] + red[(p.COT + tid) * 2 + 1]) +
                         (red[(2 * p.COT + tid) * 2 + 1] + red[(3 * p.COT + tid) * 2 + 1]);
        *reinterpret_cast<float2*>(p.stats + ((long)ptile * p.Cout + co) * 2) = make_float2(s4, q4);
      }
    }
  }
#pragma unroll
  for (int pb = 0; pb < PB; ++pb) {
    const int pix = (wave * PB + pb) * 16 + l16;
    if (pix >= npix) continue;
    const int py = pix / p.TW, px = pix - py * p.TW;
    const int oy = oy0 + py, ox = ox0 + px;
    if (oy >= p.OH || ox >= p.OW) continue;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int co = cot * p.COT + nb * 16 + kq * 4 + r;
        if (co >= p.Cout) continue;
        float v = acc[nb][pb][r];
        if (p.bias && blockIdx.z == 0) v += p.bias[co];
        float* dst;
        if constexpr (!PH) {                  // plain NCHW store
          dst = p.y + (long)b * p.outBS + (long)co * p.outCS + (long)oy * p.outRS + ox;
        } else {
          // cout' = cin*(PX*PY) + v*PX + q: x phase q (stride-(1,kw) backward-data: the kw phases of a pixel are adjacent
          // floats of dx, written by one workgroup) and/or y phase v (few-channel layers: V output rows per cout block)
          const int nph = p.outXmul * p.outYmul;
          const int cc = co / nph, phi = co - cc * nph;
          const int vph = phi / p.outXmul, q = phi - vph * p.outXmul;
          const int row = oy * p.outYmul + vph;
          if (row >= p.outH) continue;
          dst = p.y + (long)b * p.outBS + (long)cc * p.outCS + (long)row * p.outRS + (long)ox * p.outXmul + q;
        }
        if (split) atomicAdd(dst, v);         // channel slices accumulate into the zeroed output; activation follows
        else *dst = mpa_apply_act(v, p.act, p.slope);
      }
    }
  }
}

template <int NB, int PB, int KW, bool EF, bool PH = false>
int launch_fwd_ef(const FwdPlan& pl, const ConvFwdParams& p, dim3 grid, hipStream_t s) {
  static bool big_lds = false;
  if (!big_lds) {
    (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<NB, PB, KW, EF, PH>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              80 * 1024);
    big_lds = true;
  }
  MPA_LAUNCH((conv_fwd_kernel<NB, PB, KW, EF, PH>), grid, dim3(256), pl.lds_bytes, s, p);
  return mpa_launch_status();
}

template <int NB, int PB, int KW>
int launch_fwd_one(const FwdPlan& pl, const ConvFwdParams& p, dim3 grid, hipStream_t s) {
  // no edge-fix build for the 15x15 / 9x9 specialisations: they serve widths 216 and 108, and their register budget is
  // tight
  if (p.outCdiv < p.Cout) {               // phase stores: built for the generic and the 15-tap loops only (plan_fwd)
    if constexpr (KW == 0 || KW == 15) return launch_fwd_ef<NB, PB, KW, false, true>(pl, p, grid, s);
    return MPA_ERR_UNSUPPORTED;
  }
  if constexpr (KW < 9) {
    if (p.quad && (p.W & 3)) return launch_fwd_ef<NB, PB, KW, true>(pl, p, grid, s);
  }
  return launch_fwd_ef<NB, PB, KW, false>(pl, p, grid, s);
}

template <int NB, int PB>
int launch_fwd_kw(const FwdPlan& pl, const ConvFwdParams& p, dim3 grid, hipStream_t s) {
  if constexpr (PB >= 4 && (NB <= 2 || NB * PB <= 30)) {
    if (pl.KWS == 15) return launch_fwd_one<NB, PB, 15>(pl, p, grid, s);
    if (pl.KWS == 9) return launch_fwd_one<NB, PB, 9>(pl, p, grid, s);
    if (pl.KWS == 5) return launch_fwd_one<NB, PB, 5>(pl, p, grid, s);
    if (pl.KWS == 3) return launch_fwd_one<NB, PB, 3>(pl, p, grid, s);
  }
  if (pl.KWS != 0) return MPA_ERR_UNSUPPORTED;
  return launch_fwd_one<NB, PB, 0>(pl, p, grid, s);
}

template <int NB>
int launch_fwd_nb(const FwdPlan& pl, const ConvFwdParams& p, hipStream_t s) {
  dim3 grid((unsigned)(mpa_cdiv(p.nTilesAll, 8) * 8 * pl.coTiles), 1, (unsigned)mpa_cdiv(pl.nChunks, p.chunksPer));
  switch (pl.PB) {
    case 1: return launch_fwd_kw<NB, 1>(pl, p, grid, s);
    case 2: return launch_fwd_kw<NB, 2>(pl, p, grid, s);
    case 4: return launch_fwd_kw<NB, 4>(pl, p, grid, s);
    case 6: return launch_fwd_kw<NB, 6>(pl, p, grid, s);
    case 8:
      if constexpr (NB <= 4) return launch_fwd_kw<NB, 8>(pl, p, grid, s);
      return MPA_ERR_UNSUPPORTED;
    case 12:
      if constexpr (NB <= 2) return launch_fwd_kw<NB, 12>(pl, p, grid, s);
      return MPA_ERR_UNSUPPORTED;
    default: return MPA_ERR_UNSUPPORTED;
  }
}

int launch_fwd_nb3(const FwdPlan& pl, const ConvFwdParams& p, hipStream_t s) {      // phase-store builds only
  dim3 grid((unsigned)(mpa_cdiv(p.nTilesAll, 8) * 8 * pl.coTiles), 1, (unsigned)mpa_cdiv(pl.nChunks, p.chunksPer));
  if (pl.KWS != 0 || p.outCdiv >= p.Cout) return MPA_ERR_UNSUPPORTED;
  switch (pl.PB) {
    case 4: return launch_fwd_ef<3, 4, 0, false, true>(pl, p, grid, s);
    case 6: return launch_fwd_ef<3, 6, 0, false, true>(pl, p, grid, s);
    case 8: return launch_fwd_ef<3, 8, 0, false, true>(pl, p, grid, s);
    default: return MPA_ERR_UNSUPPORTED;
  }
}

int launch_fwd_nb6(const FwdPlan& pl, const ConvFwdParams& p, hipStream_t s) {      // 96-cout phase tiles
  dim3 grid((unsigned)(mpa_cdiv(p.nTilesAll, 8) * 8 * pl.coTiles), 1, (unsigned)mpa_cdiv(pl.nChunks, p.chunksPer));
  if (pl.KWS != 0 || p.outCdiv >= p.Cout || pl.PB != 4) return MPA_ERR_UNSUPPORTED;
  return launch_fwd_ef<6, 4, 0, false, true>(pl, p, grid, s);
}

int launch_fwd(const FwdPlan& pl, const ConvFwdParams& p, hipStream_t s) {
  switch (pl.NB) {
    case 3: return launch_fwd_nb3(pl, p, s);
    case 6: return launch_fwd_nb6(pl, p, s);
    case 1: return launch_fwd_nb<1>(pl, p, s);
    case 2: return launch_fwd_nb<2>(pl, p, s);
    case 4: return launch_fwd_nb<4>(pl, p, s);
    case 5: return launch_fwd_nb<5>(pl, p, s);
    default: return MPA_ERR_UNSUPPORTED;
  }
}

// ------------------------------------------------------------------------------------------------ filter packing
// packed[cot][chunk][dy][dx][ck][COTP]
//   mode 0 (forward)      : value = w[co][ci][dy][dx]
//   mode 1 (backward-data): the derived conv has Cin' = Cout, Cout' = Cin (stride 1) or kw*Cin (stride == kernel
//                           along W), value = w[ci'][co' % Cin][kh-1-dy][dxsel]
struct PackParams {
  const float* w;
  float* wp;
  int Cout_w, Cin_w, kh_w, kw_w;   // original filter dims
  int mode, xphase;                // xphase: strided-W backward (dx taken from co' / Cin)
  int yphase;                      // V > 1: cout'' = cin*V + v with the flipped filter shifted down by v rows
  int CinP, CoutP, kh, kw;         // dims of the conv that will consume the packed filters
  int CK, nChunks, COT, COTP, coTiles;
  int KWP;                         // > 0: tap-vector layout packed[cot][chunk][dy][ck][COT][KWP] (see fwd_kw_special)
  long total;
};

__device__ __forceinline__ void conv_pack_range(const PackParams& p, long first, long step) {
  for (long i = first; i < p.total; i += step) {
    long r = i;
    int col, ck, dx;
    if (p.KWP > 0) {
      int pos = (int)(r % p.KWP); r /= p.KWP;
      col = (int)(r % p.COTP); r /= p.COTP;
      ck = (int)(r % p.CK); r /= p.CK;
      // 16-tap rows: chunk g of the row sits at position (g + (col&15)>>2) & 3 -- undo the rotation to find the tap
      if (p.KWP == 16) pos = ((((pos >> 2) - ((col & 15) >> 2)) & 3) << 2) | (pos & 3);
      dx = pos;
    } else {
      col = (int)(r % p.COTP); r /= p.COTP;
      ck = (int)(r % p.CK); r /= p.CK;
      dx = (int)(r % p.kw); r /= p.kw;
    }
    const int dy = (int)(r % p.kh); r /= p.kh;
    const int chunk = (int)(r % p.nChunks); r /= p.nChunks;
    const int cot = (int)r;
    const int co = cot * p.COT + col, ci = chunk * p.CK + ck;
    float v = 0.f;
    if (col < p.COT && co < p.CoutP && ci < p.CinP && dx < p.kw) {
      if (p.mode == 0) {
        v = p.w[(((long)co * p.Cin_w + ci) * p.kh_w + dy) * p.kw_w + dx];
      } else if (p.yphase > 1) {
        const int cc = co / p.yphase, dyo = dy - (co - cc * p.yphase);
        if (dyo >= 0 && dyo < p.kh_w)
          v = p.w[(((long)ci * p.Cin_w + cc) * p.kh_w + (p.kh_w - 1 - dyo)) * p.kw_w + (p.kw_w - 1 - dx)];
      } else if (!p.xphase) {
        v = p.w[(((long)ci * p.Cin_w + co) * p.kh_w + (p.kh_w - 1 - dy)) * p.kw_w + (p.kw_w - 1 - dx)];
      } else {
        const int cc = co / p.kw_w, q = co - cc * p.kw_w;    // cout' = cin*kw + dx phase (see the epilogue)
        v = p.w[(((long)ci * p.Cin_w + cc) * p.kh_w + (p.kh_w - 1 - dy)) * p.kw_w + q];
      }
    }
    p.wp[i] = v;
  }
}

__global__ void conv_pack_kernel(const PackParams p) {
  conv_pack_range(p, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// every filter bank of a model in one launch: blockIdx.y = entry of a device-resident table of PackParams (the training
// step re-packs ~44 banks after each optimizer step; as 44 launches that was 0.21 ms of a 28 ms step at local batch 32)
__global__ void conv_pack_many_kernel(const PackParams* __restrict__ table) {
  const PackParams p = table[blockIdx.y];
  conv_pack_range(p, blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

// derived problem for backward-data
struct BwdDataGeom {
  bool ok, xphase;
  int yphase;                            // V > 1: V output rows per derived cout block (see below)
  int Cin, H, W, Cout, kh, kw, ph, pw;   // conv consuming dy (B,Cin=Cout_orig,H=OH,W=OW), stride (sh,1)
  int sh, Hplan;                         // vertical stride V and the input height the planner must assume so that the
                                         // derived conv has ceil(H_orig/V) output rows (rows past H are zero-filled)
};

// Stride-1 layers with very few input channels (the first conv: 6 HCQT harmonics) waste most of the 16-row MFMA tile in
// backward-data (6 of 16 couts).  There the derived conv computes V vertically adjacent output rows at once:
// cout'' = cin*V + v, kernel height kh+V-1 with the flipped filter shifted down by v rows, vertical stride V -- 12 of 16
// rows busy for 16/15 of the taps (backward-data of inc.double_conv.0: 3.6 -> ~2 ms).
inline int bwd_data_yphase(int Cin, int kh) {
  if (kh < 5 || Cin > 8) return 1;
  return (Cin <= 4 && kh >= 9) ? 4 : 2;
}

BwdDataGeom bwd_data_geom(const mpa_conv_desc* d) {
  BwdDataGeom g{};
  const int OH = (d->H + 2 * d->ph - d->kh) / d->sh + 1, OW = (d->W + 2 * d->pw - d->kw) / d->sw + 1;
  g.Cin = d->Cout; g.H = OH; g.W = OW; g.yphase = 1; g.sh = 1; g.Hplan = OH;
  if (d->sh == 1 && d->sw == 1) {
    g.ok = true; g.xphase = false;
    g.Cout = d->Cin; g.kh = d->kh; g.kw = d->kw; g.ph = d->kh - 1 - d->ph; g.pw = d->kw - 1 - d->pw;
    const int V = bwd_data_yphase(d->Cin, d->kh);
    if (V > 1) {
      g.yphase = V; g.sh = V; g.Cout = V * d->Cin; g.kh = d->kh + V - 1; g.Hplan = OH + V - 1;
    }
  } else if (d->sh == 1 && d->sw == d->kw && d->pw == 0 && OW * d->sw == d->W) {
    g.ok = true; g.xphase = true;       // non-overlapping windows along W: kw independent (kh x 1) convs
    g.Cout = d->kw * d->Cin; g.kh = d->kh; g.kw = 1; g.ph = d->kh - 1 - d->ph; g.pw = 0;
  } else {
    g.ok = false;
  }
  return g;
}

inline FwdPlan plan_bwd_data(const mpa_conv_desc* d, const BwdDataGeom& g) {
  return plan_fwd(d->B, g.Cin, g.Hplan, g.W, g.Cout, g.kh, g.kw, g.sh, 1, g.ph, g.pw, true, g.xphase || g.yphase > 1,
                  g.xphase ? d->sw : 1);
}

// ------------------------------------------------------------------------------------------------ backward-weight
constexpr int WGG_DEPTH = 2;      // conv_wgrad_g_kernel: 16-pixel groups whose dY quads are in flight
constexpr int WGG_SLACK = 64;      // ... and zeroed LDS words behind its X tile

struct WgPlan {
  int NBC, NTW, COT, coTiles, nPerBlock, nTiles, Ntot, XCH, TH, TW, DP, tilesY, tilesX, IH, IW, LW, XCHP, DCP, S, OH, OW;
  size_t lds_bytes;
  bool ok;
  int quad, xshift;   // 16-byte LDS-DMA staging: 4-aligned window origin (x0a = ix0 - xshift), pitches % 4 == 0
  int ef;             // row ends are not quad aligned: edge_fix_* completes the straddling quads
  int ga;             // conv_wgrad_g_kernel: dY operand from global memory, LDS holds the X tile only
};

WgPlan plan_wgrad(const mpa_conv_desc* d) {
  WgPlan best{};
  best.ok = false;
  const int OH = (d->H + 2 * d->ph - d->kh) / d->sh + 1, OW = (d->W + 2 * d->pw - d->kw) / d->sw + 1;
  if (OH <= 0 || OW <= 0) return best;
  const int khkw = d->kh * d->kw;
  const int Ntot = d->Cin * khkw;
  // wave tile variants (cout blocks x tap blocks); many tap blocks per wave = few dY floats staged per MFMA
  const int var_nbc[5] = {1, 2, 2, 4, 5}, var_ntw[5] = {16, 8, 16, 6, 6};
  double bestcost = 1e300;
  const char* var_env = getenv("MPA_WG_VARIANT");        // diagnostics / tests: restrict the search to one wave tile
  for (int v = 0; v < 5; ++v) {
    if (var_env && atoi(var_env) != v) continue;
    WgPlan pl{};
    pl.OH = OH; pl.OW = OW; pl.Ntot = Ntot;
    pl.NBC = var_nbc[v]; pl.NTW = var_ntw[v];
    pl.COT = pl.NBC * 16;
    pl.coTiles = (int)mpa_cdiv(d->Cout, pl.COT);
    pl.nPerBlock = 4 * pl.NTW * 16;
    pl.nTiles = (int)mpa_cdiv(Ntot, pl.nPerBlock);
    pl.XCH = std::min(d->Cin, (pl.nPerBlock + khkw - 2) / khkw + 1);
    // cost of a variant = cost of one (pixel tile, block) x the number of (cout tile, tap tile) blocks that have to visit
    // every pixel tile.  (Round 1 multiplied by the padding ratio only, which compared the *per-block* cost of variants
    // whose blocks cover different amounts of work: for 128->200 3x3 it picked 7 x 3 blocks of <2,8> over 3 x 3 of <5,6>.)
    static const bool old_norm = getenv("MPA_WG_COSTNORM") && atoi(getenv("MPA_WG_COSTNORM")) == 0;   // diagnostics
    const double pad_eff = old_norm ? ((double)pl.coTiles * pl.COT / d->Cout) * ((double)pl.nTiles * pl.nPerBlock / Ntot)
                                    : (double)pl.coTiles * pl.nTiles / 8.0;
    static const int force_txn = getenv("MPA_WG_TXN") ? atoi(getenv("MPA_WG_TXN")) : 0;   // diagnostics
    // dY-from-global variant: exact tiling of 4-aligned rows, stride 1 (any variant) or the head's stride 3 (<5,6>)
    const char* ga_env = getenv("MPA_WG_GA");            // diagnostics / tests: "0" = never, "force" = whenever feasible
    const bool no_ga = ga_env && ga_env[0] == '0', force_ga = ga_env && ga_env[0] == 'f';
    bool ga_found = false;
    const bool ga_sw = d->sw == 1 || (d->sw == 3 && pl.NBC == 5);
    const bool ga_ef = (OW & 3) || (d->W & 3);      // unaligned rows: one tile per row, <2,8> only, stride 1
    if (!no_ga && !force_txn && ga_sw && (!ga_ef || (pl.NBC == 2 && pl.NTW == 8 && d->sw == 1))) {
      for (int txn = 1; txn <= (ga_ef ? 1 : 16); ++txn) {
        if (OW % txn) continue;
        const int TW = OW / txn;
        if ((!ga_ef && (TW & 3)) || TW < 16) continue;
        const int DPg = (int)mpa_cdiv(TW, 4) * 4;
        const int IW = (DPg - 1) * d->sw + d->kw;
        const int xshift = ((-d->pw) % 4 + 4) % 4;
        const int LW = (int)mpa_cdiv(IW + 3, 4) * 4;
        for (int THmax = std::min(OH, 64); THmax >= 1; --THmax) {
          if ((mpa_cdiv((long)pl.XCH * ((THmax - 1) * d->sh + d->kh) * LW, 64) * 64 + WGG_SLACK) * 4 > 64 * 1024) continue;
          const int ty = (int)mpa_cdiv(OH, THmax);
          const int TH = (int)mpa_cdiv(OH, ty);        // the largest tile that fits, then balanced over the rows
          const int IH = (TH - 1) * d->sh + d->kh;
          const int XCHP = IH * LW;
          const long floats = mpa_cdiv((long)pl.XCH * XCHP, 64) * 64 + WGG_SLACK;
          if (ga_ef && (long)pl.XCH * IH > 256 * EDGE_MAXF) continue;
          const int ksteps = (TW >> 4) * 4 + ((TW & 15) > 12 ? 4 : (TW & 15) ? 3 : 0);
          // per k-step: the MFMAs + one B read per tap block (+ its address add once per group); dY costs nothing here
          const double mfma = (double)TH * (ksteps * (pl.NBC * pl.NTW * 32.0 + 5.0 * pl.NTW) + 120.0);
          const double words = (double)pl.XCH * IH * LW;
          const double stage = words / 256.0 * 80.0;
          const double cost = (double)ty * txn * (mfma + 0.7 * stage + 600.0) * pad_eff;
          const bool take = force_ga ? (!best.ga || cost < bestcost) : cost < bestcost;
          if (take) {
            ga_found = true;
            bestcost = cost;
            best = pl;
            best.TH = TH; best.TW = TW; best.DP = DPg; best.tilesY = ty; best.tilesX = txn; best.IH = IH; best.IW = IW;
            best.LW = LW; best.XCHP = XCHP; best.DCP = 0; best.lds_bytes = (size_t)floats * 4; best.ok = true;
            best.quad = 1; best.xshift = xshift; best.ef = ga_ef ? 1 : 0; best.ga = 1;
          }
          break;
        }
      }
    }
    if (force_ga && (ga_found || best.ga)) continue;      // a dY-from-global plan exists: skip the LDS-staged candidates
    for (int txn = 1; txn <= std::min(OW, 64); ++txn) {
      if (force_txn && txn != std::min(force_txn, OW)) continue;
      int TW = (int)mpa_cdiv(OW, txn);
      const int DP = (int)mpa_cdiv(TW, 4) * 4;
      const int IW = (DP - 1) * d->sw + d->kw;
      // 16-byte LDS-DMA staging: stride 1 and tile origins on multiples of 4 (the tile width is rounded up for that;
      // the global side needs no alignment).  Row ends that are not quad aligned are completed by edge_fix_*.
      const int quad = 1;      // the staged windows are contiguous column blocks whatever the stride of the taps
      if (quad && txn > 1) TW = DP;
      const int tilesX = (int)mpa_cdiv(OW, TW);
      if (quad && tilesX != txn) continue;                  // the same tiling is reached from a smaller txn
      const int ef = quad && ((d->W & 3) || (OW & 3)) ? 1 : 0;
      // window origin ix0 = ox0*sw - pw with ox0 a multiple of 4: its misalignment is the same for every tile
      const int xshift = quad ? ((-d->pw) % 4 + 4) % 4 : 0;
      const int LW = quad ? (int)mpa_cdiv(IW + 3, 4) * 4 : (IW | 1);
      for (int TH = std::min(OH, 64); TH >= 1; --TH) {
        const int IH = (TH - 1) * d->sh + d->kh;
        const int XCHP = IH * LW;
        const int DCP = round_mod(TH * DP, 32, quad ? 4 : 2);
        const long floats = mpa_cdiv((long)pl.XCH * XCHP, 64) * 64 + mpa_cdiv((long)pl.COT * DCP, 64) * 64;
        if (floats * 4 > 64 * 1024) continue;
        if (ef && ((long)pl.XCH * IH > 256 * EDGE_MAXF || (long)pl.COT * TH > 256 * EDGE_MAXF)) continue;
        const int ty = (int)mpa_cdiv(OH, TH);
        // cycles per tile: MFMA issue (per wave) + staging.  An LDS-DMA wave instruction costs the CU ~80 cycles
        // whatever its width: 64 words (dword form) or 256 words (16-byte form) each, four waves issuing in turn.
        // per k-step: NBC*NTW MFMAs of 32 cycles plus ~2 non-MFMA vector instructions (read + address) per operand at
        // ~4 cycles each; per tile row: ~30 instructions of loop set-up (measured: 37x4 tiles ran 26 % slower than 8x36)
        // (rows of at least 16 pixels run the unrolled loop with immediate offsets: ~1 instead of ~2 such instructions)
        const double opi = (d->sw == 1 && DP >= 16) ? 4.0 : 8.0;
        const double mfma = (double)TH * ((DP / 4) * (pl.NBC * pl.NTW * 32.0 + opi * (pl.NBC + pl.NTW)) + 120.0);
        const double words = (double)pl.XCH * IH * LW + (double)pl.COT * TH * DP;
        const double stage = words / (quad ? 256.0 : 64.0) * 80.0;
        const double cost = (double)ty * txn * (mfma + 0.7 * stage + 600.0) * pad_eff;
        if (cost < bestcost) {
          bestcost = cost;
          best = pl;
          best.TH = TH; best.TW = TW; best.DP = DP; best.tilesY = ty; best.tilesX = txn; best.IH = IH; best.IW = IW;
          best.LW = LW; best.XCHP = XCHP; best.DCP = DCP; best.lds_bytes = (size_t)floats * 4; best.ok = true;
          best.quad = quad; best.xshift = xshift; best.ef = ef;
        }
        break;   // largest TH that fits for this TW
      }
    }
  }
  if (!best.ok) return best;
  // split the (image, tile) loop over S blocks so that the grid is a whole number of resident waves of workgroups
  const long totalTiles = (long)d->B * best.tilesY * best.tilesX;
  const long per_cu = std::max<long>(1, std::min<long>(2, (160 * 1024) / (long)best.lds_bytes));
  const long slots = 256 * per_cu;
  const long groups = (long)best.nTiles * best.coTiles;
  long S = std::max<long>(1, (2 * slots) / groups);
  if (groups * S < slots && S < totalTiles) S = mpa_cdiv(slots, groups);
  if (S > totalTiles) S = totalTiles;
  if (S > 1024) S = 1024;
  // few tiles per slice: pick the slice count by the same small model as plan_wgrad15 (the busiest CU's workgroups x
  // tiles per slice x tile time, + 8 % when a CU holds a single workgroup, + one write and read of the partial sums
  // per slice) -- 160 tiles over 93 slices are 2 tiles for most workgroups and 1 for the rest
  if (totalTiles / S < 8 && !getenv("MPA_WG_S_OLD")) {
    const long lo = std::max<long>(1, S / 2), hi = std::min<long>(std::min<long>(2 * S, totalTiles), 1024);
    const double t_tile = 1.1 * (double)best.TH * (best.DP / 4) * best.NBC * best.NTW * 32.0 / 2.4e9;
    const double t_slice = (double)std::min(d->Cout, best.COT * best.coTiles) * (best.Ntot + 1) * 8.0 / 4.0e12;
    double bestc = 1e300;
    long bestS = S;
    for (long c = lo; c <= hi; ++c) {
      const long per_cu_wgs = mpa_cdiv(c * groups, 256);
      const double est = (double)per_cu_wgs * (double)mpa_cdiv(totalTiles, c) * t_tile * (per_cu_wgs < 2 ? 1.08 : 1.0) +
                         (double)c * t_slice;
      if (est < bestc) { bestc = est; bestS = c; }
    }
    S = bestS;
  }
  if (const char* e = getenv("MPA_WG_S")) {      // diagnostics: force the slice count
    const long f = atol(e);
    if (f >= 1 && f <= std::min<long>(totalTiles, 1024)) S = f;
  }
  best.S = (int)S;
  return best;
}

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


// ------------------------------------------------------------------------------------------------ backward-weight, 15x15
// 88 % of the model's conv FLOPs sit in 15x15 stride-1 filters (inc, down1, upconv4; DRCNN prefilters), so their
// weight gradient gets a dedicated kernel: the MFMA N dimension is the 15 dx taps (padded to 16) of one (ci, dy) row
// and the X tile has a fixed LDS row pitch of 128 words, so every B-operand read is `base + immediate` (dy*512 B)
// -- one address VGPR for 15 reads instead of a running pointer per tap block.
//   wave tile: NBC cout blocks x CIW input channels x 15 dy  (acc = NBC*CIW*15 tiles of 16x16)
//   block    : 4 waves = 4*CIW input channels sharing one dY tile of NBC*16 couts
constexpr int W15_PITCH = 128;
constexpr int W15G_DEPTH = 2;      // dY-from-global variant: groups of 16 pixels whose dY quads are in flight

struct Wg15Params {
  const float* x;
  const float* dy;
  float* ws;
  int B, Cin, H, W, Cout, OH, OW;
  int COT, TH, TW, DP, tilesY, tilesX, IH, IW, DCP, S, Ntot, TX64, TD64;
  int quad;  // 16-byte LDS-DMA staging (aligned geometry): the X window then starts one column further left (ox0-8)
  int co_base;   // first cout of this launch's cout tiles (remainder launch: couts past the last full NBC = 2 tile)
  int dbg;   // diagnostics (env MPA_DEBUG_WG15): 1 = stage only the first tile, 2 = skip the MFMA loop, 3 = 1 + no barriers
  int fold_R;   // conv_wgrad15f_kernel: the couts [co_base, co_base + fold_R) of this launch, fold_R <= 16 / ceil(15 / NT)
};

template <int NBC, int CIW>
__global__ __launch_bounds__(256) void conv_wgrad15_kernel(const Wg15Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_x = lds;
  float* lds_dy = lds + p.TX64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, l16 = lane & 15;
  const int split = blockIdx.x, cig = blockIdx.y, cot = blockIdx.z;
  const int ci_first = cig * 4 * CIW;
  const int xchp = p.IH * W15_PITCH;
  // (Two workgroups share a CU and start in lock-step.  Giving the one whose LDS allocation does not start at 0 issue
  // priority, so that each one's staging falls into the other's compute phase, was measured in round 1 and changed
  // nothing: the LDS-DMA issue cost is paid by the CU whatever the phase between the two -- see conv_wgrad15g_kernel.)
  const int NtotP = p.Ntot + 1;
  const bool do_bias = cig == 0;
  float bsum = 0.f;
  f32x4 acc[NBC][CIW][15];
#pragma unroll
  for (int a = 0; a < NBC; ++a)
#pragma unroll
    for (int c = 0; c < CIW; ++c)
#pragma unroll
      for (int t = 0; t < 15; ++t) acc[a][c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int tilesPerImg = p.tilesY * p.tilesX;
  const long totalTiles = (long)p.B * tilesPerImg;
  const float* bbase = lds_x + wave * CIW * xchp + kq + l16 + (p.quad ? 1 : 0);
  const float* abase = lds_dy + l16 * p.DCP + kq;
  for (long tile = split; tile < totalTiles; tile += p.S) {
    const int b = (int)(tile / tilesPerImg);
    const int tr = (int)(tile - (long)b * tilesPerImg);
    const int ty = tr / p.tilesX, tx = tr - ty * p.tilesX;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;
    if (p.dbg != 3) __syncthreads();
    if ((p.dbg != 1 && p.dbg != 3) || tile == split) {
      if (p.quad) {
        glds_stage_x16(lds_x, p.x + (long)b * p.Cin * p.H * p.W, lane, wave, 4 * CIW, p.IH, W15_PITCH, xchp, p.TX64,
                       ci_first, oy0 - 7, ox0 - 8, p.Cin, p.H, p.W);
        glds_stage_dy16(lds_dy, p.dy + (long)b * p.Cout * p.OH * p.OW, lane, wave, p.COT, p.TH, p.DP, p.DCP, p.TD64,
                        cot * p.COT, oy0, ox0, p.Cout, p.OH, p.OW, min(p.OW, ox0 + p.TW));
      } else {
        glds_stage_x(lds_x, p.x + (long)b * p.Cin * p.H * p.W, lane, wave, 4 * CIW, p.IH, p.IW, W15_PITCH, xchp, p.TX64,
                     ci_first, oy0 - 7, ox0 - 7, p.Cin, p.H, p.W);
        glds_stage_dy(lds_dy, p.dy + (long)b * p.Cout * p.OH * p.OW, lane, wave, p.COT, p.TH, p.DP, p.DCP, p.TD64,
                      cot * p.COT, oy0, ox0, p.Cout, p.OH, p.OW, min(p.OW, ox0 + p.TW));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p.dbg != 3 || tile == split) __syncthreads();
    if (p.dbg == 2) continue;
    if (do_bias) {
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
      const float* ap = abase + py * p.DP;
      const float* bp = bbase + py * W15_PITCH;
      // two explicit operand register sets: the LDS reads of step k+1 are issued before the MFMAs of step k
      float a0[NBC], b0[CIW][15], a1[NBC], b1[CIW][15];
#define W15_LOAD(A, Bv, PX)                                                                \
  {                                                                                        \
    _Pragma("unroll") for (int cb = 0; cb < NBC; ++cb) A[cb] = ap[cb * 16 * p.DCP + (PX)]; \
    _Pragma("unroll") for (int c = 0; c < CIW; ++c)                                        \
      _Pragma("unroll") for (int t = 0; t < 15; ++t) Bv[c][t] = bp[c * xchp + t * W15_PITCH + (PX)]; \
  }
#define W15_MMA(A, Bv)                                                                      \
  {                                                                                         \
    _Pragma("unroll") for (int c = 0; c < CIW; ++c)                                         \
      _Pragma("unroll") for (int t = 0; t < 15; ++t)                                        \
        _Pragma("unroll") for (int cb = 0; cb < NBC; ++cb)                                  \
          acc[cb][c][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(A[cb], Bv[c][t], acc[cb][c][t], 0, 0, 0); \
  }
      W15_LOAD(a0, b0, 0)
      for (int px0 = 0; px0 < p.DP; px0 += 8) {
        const int p1 = px0 + 4 < p.DP ? px0 + 4 : px0;
        W15_LOAD(a1, b1, p1)
        __builtin_amdgcn_sched_barrier(0);
        W15_MMA(a0, b0)
        __builtin_amdgcn_sched_barrier(0);
        if (px0 + 4 < p.DP) {
          const int p2 = px0 + 8 < p.DP ? px0 + 8 : px0 + 4;
          W15_LOAD(a0, b0, p2)
          __builtin_amdgcn_sched_barrier(0);
          W15_MMA(a1, b1)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#undef W15_LOAD
#undef W15_MMA
    }
  }
  // D[row = cout (kq*4+r)][col = dx (l16)]
  float* out = p.ws + (long)split * p.Cout * NtotP;
  if (l16 < 15) {
#pragma unroll
    for (int c = 0; c < CIW; ++c) {
      const int ci = ci_first + wave * CIW + c;
      if (ci >= p.Cin) continue;
#pragma unroll
      for (int t = 0; t < 15; ++t)
#pragma unroll
        for (int cb = 0; cb < NBC; ++cb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = cot * p.COT + cb * 16 + kq * 4 + r;
            if (co < p.Cout) out[(long)co * NtotP + ci * 225 + t * 15 + l16] = acc[cb][c][t][r];
          }
    }
  }
  if (do_bias) {
    bsum += __shfl_xor(bsum, 1, 64);
    const int co = cot * p.COT + (tid >> 1);
    if ((tid & 1) == 0 && (tid >> 1) < p.COT && co < p.Cout) out[(long)co * NtotP + p.Ntot] = bsum;
  }
}

// Variant with the dY operand read straight from global memory (quad geometry: OW % 4 == 0, tile origins % 4 == 0).
// Staging dY through LDS costs time in proportion to its bytes that nothing hides (measured: 42 KB dY + 35 KB X per
// 3x108 tile = 7 % of the kernel, whatever the mechanism -- LDS-DMA or registers -- and whatever the phase between the
// two co-resident workgroups).  Here a lane loads one float4 = 4 consecutive pixels of its cout row and uses it as the A
// operand of 4 consecutive k-steps: k-step j of a 16-pixel group contracts pixels {16g + 4kq + j}, so the B operand
// sits at `4kq + dx + (16g + j)` -- still base + immediate.  The next group's float4 is in flight during the current
// group's 120 MFMAs.  LDS then holds the X tile only, which buys TH up to 25 rows (halo overhead 1.6x instead of 5.7x).
// A row's last DP % 16 pixels are a tail of 1-3 ordinary k-steps (pixels {4s + kq}, dword loads).
// EVEN: the number of full groups per row is even, the group loop runs two groups per trip and the rotation of the
// W15G_DEPTH = 2 in-flight dY register sets is register renaming instead of 16 v_mov per group.
template <int NBC, bool TAIL, bool EVEN = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad15g_kernel(const Wg15Params p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_x = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, l16 = lane & 15;
  const int split = blockIdx.x, cig = blockIdx.y, cot = blockIdx.z;
  const int ci_first = cig * 4;
  const int xchp = p.IH * W15_PITCH;
  const int NtotP = p.Ntot + 1;
  const bool do_bias = cig == 0 && wave == 0;
  float bs[NBC];
  f32x4 acc[NBC][15];
#pragma unroll
  for (int a = 0; a < NBC; ++a) {
    bs[a] = 0.f;
#pragma unroll
    for (int t = 0; t < 15; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const int tilesPerImg = p.tilesY * p.tilesX;
  const long totalTiles = (long)p.B * tilesPerImg;
  const int nfull = p.DP >> 4, tail = (p.DP & 15) >> 2;     // nfull >= 1; TAIL == (tail != 0)
  const float* bfull = lds_x + wave * xchp + 4 * kq + l16 + 1;
  const float* btail = lds_x + wave * xchp + kq + l16 + 1 + 16 * nfull;
  const int plane = p.OH * p.OW;
  const int loff = (l16 * plane + 4 * kq) * 4, loff_t = (l16 * plane + kq) * 4;     // bytes

  for (long tile = split; tile < totalTiles; tile += p.S) {
    const int b = (int)(tile / tilesPerImg);
    const int tr = (int)(tile - (long)b * tilesPerImg);
    const int ty = tr / p.tilesX, tx = tr - ty * p.tilesX;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;
    // dY quads by buffer loads: the wave-uniform part of the address (cout block, row, group) goes into the resource's
    // base and num_records on the scalar unit, the lane part (cout row l16, quad kq) is one VGPR for the whole kernel,
    // and whatever falls outside the image's dY -- couts past Cout, rows past OH (num_records = 0) -- reads as zero
    // without a single vector instruction.  Tiles are exact in x (planner), so there is no per-lane column test.
    const float* imgb = p.dy + (long)b * p.Cout * plane;
    const int img_elems = p.Cout * plane;
    auto dy_rsrc = [&](int cb, int py, int col, bool on) {
      const int u = (p.co_base + cot * p.COT + cb * 16) * plane + (oy0 + py) * p.OW + ox0 + col;
      const int left = (on && oy0 + py < p.OH && u < img_elems) ? (img_elems - u) * 4 : 0;
      return __builtin_amdgcn_make_buffer_rsrc((void*)(imgb + u), 0, left, 0x00020000);
    };
    auto load_full = [&](float4* a, int py, int g) {
#pragma unroll
      for (int cb = 0; cb < NBC; ++cb)
        a[cb] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(dy_rsrc(cb, py, 16 * g, true), loff, 0, 0));
    };
    __syncthreads();
    if (p.dbg != 1 || tile == split)
      glds_stage_x16(lds_x, p.x + (long)b * p.Cin * p.H * p.W, lane, wave, 4, p.IH, W15_PITCH, xchp, p.TX64, ci_first,
                     oy0 - 7, ox0 - 8, p.Cin, p.H, p.W);
    float4 an[W15G_DEPTH][NBC];       // dY quads of the next W15G_DEPTH groups, in flight
#pragma unroll
    for (int d = 0; d < W15G_DEPTH; ++d) load_full(an[d], d / nfull, d % nfull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (p.dbg == 2) continue;

    float b0[15], b1[15];
#define W15G_LOAD(Bv, BP, IMM)                                                              \
  { _Pragma("unroll") for (int t = 0; t < 15; ++t) Bv[t] = (BP)[t * W15_PITCH + (IMM)]; }
#define W15G_MMA(AV, AC, Bv)                                                                \
  {                                                                                         \
    _Pragma("unroll") for (int t = 0; t < 15; ++t)                                          \
      _Pragma("unroll") for (int cb = 0; cb < NBC; ++cb)                                    \
        acc[cb][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[cb].AC, Bv[t], acc[cb][t], 0, 0, 0); \
  }
    const float* bp = bfull;
    W15G_LOAD(b0, bp, 0)
    for (int py = 0; py < p.TH; ++py) {
      float4 at[NBC];
      if constexpr (TAIL) {     // the row's last 4..12 pixels: k-step s contracts pixels {16 nfull + 4s + kq}
#pragma unroll
        for (int cb = 0; cb < NBC; ++cb) {
          at[cb].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc(cb, py, 16 * nfull, true), loff_t, 0, 0));
          at[cb].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc(cb, py, 16 * nfull + 4, tail > 1), loff_t, 0, 0));
          at[cb].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dy_rsrc(cb, py, 16 * nfull + 8, tail > 2), loff_t, 0, 0));
          at[cb].w = 0.f;
        }
      }
      auto group = [&](const int g) {
        float4 ac[NBC];
#pragma unroll
        for (int cb = 0; cb < NBC; ++cb) {
          ac[cb] = an[0][cb];
#pragma unroll
          for (int d = 0; d + 1 < W15G_DEPTH; ++d) an[d][cb] = an[d + 1][cb];
        }
        const bool last = g + 1 == nfull;
        const int npy = last ? py + 1 : py;
        {
          int gd = g + W15G_DEPTH, pyd = py;
          while (gd >= nfull) { gd -= nfull; ++pyd; }
          if (pyd < p.TH) load_full(an[W15G_DEPTH - 1], pyd, gd);
        }
        const float* bpn = last ? (TAIL ? btail + py * W15_PITCH : bfull + npy * W15_PITCH) : bp + 16;
        if (do_bias) {
#pragma unroll
          for (int cb = 0; cb < NBC; ++cb) bs[cb] += (ac[cb].x + ac[cb].y) + (ac[cb].z + ac[cb].w);
        }
        W15G_LOAD(b1, bp, 1)
        __builtin_amdgcn_sched_barrier(0);
        W15G_MMA(ac, x, b0)
        __builtin_amdgcn_sched_barrier(0);
        W15G_LOAD(b0, bp, 2)
        __builtin_amdgcn_sched_barrier(0);
        W15G_MMA(ac, y, b1)
        __builtin_amdgcn_sched_barrier(0);
        W15G_LOAD(b1, bp, 3)
        __builtin_amdgcn_sched_barrier(0);
        W15G_MMA(ac, z, b0)
        __builtin_amdgcn_sched_barrier(0);
        W15G_LOAD(b0, bpn, 0)
        __builtin_amdgcn_sched_barrier(0);
        W15G_MMA(ac, w, b1)
        __builtin_amdgcn_sched_barrier(0);
        bp = bpn;
      };
      if constexpr (EVEN) {
        static_assert(W15G_DEPTH == 2, "the two-groups-per-trip loop renames exactly two in-flight sets");
        for (int g = 0; g < nfull; g += 2) { group(g); group(g + 1); }
      } else {
        for (int g = 0; g < nfull; ++g) group(g);
      }
      if constexpr (TAIL) {
        // always three k-steps: those past `tail` have A == 0 (the planner prefers DP % 16 in {0, 12})
        const float* bpn = bfull + (py + 1) * W15_PITCH;
        if (do_bias) {
#pragma unroll
          for (int cb = 0; cb < NBC; ++cb) bs[cb] += (at[cb].x + at[cb].y) + at[cb].z;
        }
        W15G_LOAD(b1, bp, 4)
        __builtin_amdgcn_sched_barrier(0);
        W15G_MMA(at, x, b0)
        __builtin_amdgcn_sched_barrier(0);
        W15G_LOAD(b0, bp, 8)
        __builtin_amdgcn_sched_barrier(0);
        W15G_MMA(at, y, b1)
        __builtin_amdgcn_sched_barrier(0);
        W15G_LOAD(b1, bpn, 0)
        __builtin_amdgcn_sched_barrier(0);
        W15G_MMA(at, z, b0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 15; ++t) b0[t] = b1[t];
        bp = bpn;
      }
    }
#undef W15G_LOAD
#undef W15G_MMA
  }
  // D[row = cout (kq*4+r)][col = dx (l16)]
  float* out = p.ws + (long)split * p.Cout * NtotP;
  const int ci = ci_first + wave;
  if (l16 < 15 && ci < p.Cin) {
#pragma unroll
    for (int t = 0; t < 15; ++t)
#pragma unroll
      for (int cb = 0; cb < NBC; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = p.co_base + cot * p.COT + cb * 16 + kq * 4 + r;
          if (co < p.Cout) out[(long)co * NtotP + ci * 225 + t * 15 + l16] = acc[cb][t][r];
        }
  }
  if (do_bias) {
#pragma unroll
    for (int cb = 0; cb < NBC; ++cb) {
      float v = bs[cb];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int co = p.co_base + cot * p.COT + cb * 16 + l16;
      if (kq == 0 && co < p.Cout) out[(long)co * NtotP + p.Ntot] = v;
    }
  }
}

// Tap-folded variant of conv_wgrad15g_kernel for a remainder of R <= 8 couts (70 = 4 x 16 + 6: DRCNN:L's prefilters;
// 20 = 16 + 4, 40 = 32 + 8, 100 = 96 + 4; the 6 and 8 couts of the test configurations).  A 16-row MFMA tile with R
// real cout rows wastes 16 - R of them.  Here the 16 rows are FS = ceil(15 / NT) copies of the R couts (NT = 8: two
// copies, R <= 8; NT = 4: four copies, R <= 4), copy s reading its dY  s NT rows *above* copy 0's:
//     acc[(c, s)][t][dx] = sum_{v, px} dY[c][v - s NT][px] * X[v + t - 7][px + dx - 7]  =  dW[c][t + s NT][dx]
// so NT tap rows of MFMAs per pixel step cover all 15 (NT / 15 of the MFMA work), a tile needs TH + NT - 1 rows of X
// instead of TH + 14 (taller tiles fit the same LDS), and the price is that the tiles cover OH + (FS - 1) NT "virtual"
// rows v.  The dY row depends on the lane, so the buffer offset and its bounds test (row in [0, OH), copy < FS) are
// per-lane VALU work: ~5 instructions per 16-pixel group against 4 NT MFMAs.  Everything else -- X tile by LDS-DMA,
// dY quads in flight W15G_DEPTH groups ahead, TAIL / EVEN -- is conv_wgrad15g_kernel's.
template <int NT, bool TAIL, bool EVEN>
__global__ __launch_bounds__(256, 2) void conv_wgrad15f_kernel(const Wg15Params p) {
  constexpr int FS = (15 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_x = lds;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int kq = lane >> 4, l16 = lane & 15;
  const int split = blockIdx.x, cig = blockIdx.y;
  const int ci_first = cig * 4;
  const int xchp = p.IH * W15_PITCH;
  const int NtotP = p.Ntot + 1;
  const bool do_bias = cig == 0 && wave == 0;
  const int R = p.fold_R;
  const int fs = l16 / R, fc = l16 - fs * R;     // this lane's A row: cout co_base + fc, copy fs
  const bool lane_on = fs < FS;
  float bs = 0.f;
  f32x4 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int tilesPerImg = p.tilesY * p.tilesX;
  const long totalTiles = (long)p.B * tilesPerImg;
  const int nfull = p.DP >> 4, tail = (p.DP & 15) >> 2;     // nfull >= 1; TAIL == (tail != 0)
  const float* bfull = lds_x + wave * xchp + 4 * kq + l16 + 1;
  const float* btail = lds_x + wave * xchp + kq + l16 + 1 + 16 * nfull;
  const int plane = p.OH * p.OW;
  const int lane_c = (fc * plane + 4 * kq) * 4, lane_ct = (fc * plane + kq) * 4;     // bytes
  constexpr int OUTSIDE = 0x7FFFFFF0;      // past num_records: the buffer load returns zero

  for (long tile = split; tile < totalTiles; tile += p.S) {
    const int b = (int)(tile / tilesPerImg);
    const int tr = (int)(tile - (long)b * tilesPerImg);
    const int ty = tr / p.tilesX, tx = tr - ty * p.tilesX;
    const int oy0 = ty * p.TH, ox0 = tx * p.TW;
    // one buffer resource per image: the R cout planes of this launch; row and column go into the lane's offset
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dy + ((long)b * p.Cout + p.co_base) * plane), 0,
                                                        R * plane * 4, 0x00020000);
    const int row0 = oy0 - fs * NT;      // this lane's dY row at py = 0
    auto voff = [&](int py, int col, int lc, bool on) {
      const int row = row0 + py;
      const bool ok = lane_on && on && (unsigned)row < (unsigned)p.OH;
      return ok ? lc + (row * p.OW + ox0 + col) * 4 : OUTSIDE;
    };
    auto load_full = [&](float4& a, int py, int g) {
      a = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff(py, 16 * g, lane_c, true), 0, 0));
    };
    __syncthreads();
    glds_stage_x16(lds_x, p.x + (long)b * p.Cin * p.H * p.W, lane, wave, 4, p.IH, W15_PITCH, xchp, p.TX64, ci_first,
                   oy0 - 7, ox0 - 8, p.Cin, p.H, p.W);
    float4 an[W15G_DEPTH];       // dY quads of the next W15G_DEPTH groups, in flight
#pragma unroll
    for (int d = 0; d < W15G_DEPTH; ++d) load_full(an[d], d / nfull, d % nfull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    float b0[NT], b1[NT];
#define W15F_LOAD(Bv, BP, IMM)                                                              \
  { _Pragma("unroll") for (int t = 0; t < NT; ++t) Bv[t] = (BP)[t * W15_PITCH + (IMM)]; }
#define W15F_MMA(AV, AC, Bv)                                                                \
  { _Pragma("unroll") for (int t = 0; t < NT; ++t)                                          \
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV.AC, Bv[t], acc[t], 0, 0, 0); }
    const float* bp = bfull;
    W15F_LOAD(b0, bp, 0)
    for (int py = 0; py < p.TH; ++py) {
      float4 at;
      if constexpr (TAIL) {     // the row's last 4..12 pixels: k-step s contracts pixels {16 nfull + 4s + kq}
        at.x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff(py, 16 * nfull, lane_ct, true), 0, 0));
        at.y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff(py, 16 * nfull + 4, lane_ct, tail > 1), 0, 0));
        at.z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff(py, 16 * nfull + 8, lane_ct, tail > 2), 0, 0));
        at.w = 0.f;
      }
      auto group = [&](const int g) {
        const float4 ac = an[0];
#pragma unroll
        for (int d = 0; d + 1 < W15G_DEPTH; ++d) an[d] = an[d + 1];
        const bool last = g + 1 == nfull;
        const int npy = last ? py + 1 : py;
        {
          int gd = g + W15G_DEPTH, pyd = py;
          while (gd >= nfull) { gd -= nfull; ++pyd; }
          if (pyd < p.TH) load_full(an[W15G_DEPTH - 1], pyd, gd);
        }
        const float* bpn = last ? (TAIL ? btail + py * W15_PITCH : bfull + npy * W15_PITCH) : bp + 16;
        if (do_bias) bs += (ac.x + ac.y) + (ac.z + ac.w);
        W15F_LOAD(b1, bp, 1)
        __builtin_amdgcn_sched_barrier(0);
        W15F_MMA(ac, x, b0)
        __builtin_amdgcn_sched_barrier(0);
        W15F_LOAD(b0, bp, 2)
        __builtin_amdgcn_sched_barrier(0);
        W15F_MMA(ac, y, b1)
        __builtin_amdgcn_sched_barrier(0);
        W15F_LOAD(b1, bp, 3)
        __builtin_amdgcn_sched_barrier(0);
        W15F_MMA(ac, z, b0)
        __builtin_amdgcn_sched_barrier(0);
        W15F_LOAD(b0, bpn, 0)
        __builtin_amdgcn_sched_barrier(0);
        W15F_MMA(ac, w, b1)
        __builtin_amdgcn_sched_barrier(0);
        bp = bpn;
      };
      if constexpr (EVEN) {
        static_assert(W15G_DEPTH == 2, "the two-groups-per-trip loop renames exactly two in-flight sets");
        for (int g = 0; g < nfull; g += 2) { group(g); group(g + 1); }
      } else {
        for (int g = 0; g < nfull; ++g) group(g);
      }
      if constexpr (TAIL) {
        const float* bpn = bfull + (py + 1) * W15_PITCH;
        if (do_bias) bs += (at.x + at.y) + at.z;
        W15F_LOAD(b1, bp, 4)
        __builtin_amdgcn_sched_barrier(0);
        W15F_MMA(at, x, b0)
        __builtin_amdgcn_sched_barrier(0);
        W15F_LOAD(b0, bp, 8)
        __builtin_amdgcn_sched_barrier(0);
        W15F_MMA(at, y, b1)
        __builtin_amdgcn_sched_barrier(0);
        W15F_LOAD(b1, bpn, 0)
        __builtin_amdgcn_sched_barrier(0);
        W15F_MMA(at, z, b0)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) b0[t] = b1[t];
        bp = bpn;
      }
    }
#undef W15F_LOAD
#undef W15F_MMA
  }
  // D[row = (copy, cout) (kq*4+r)][col = dx (l16)]
  float* out = p.ws + (long)split * p.Cout * NtotP;
  const int ci = ci_first + wave;
  if (l16 < 15 && ci < p.Cin) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = kq * 4 + r, s = m / R, c = m - s * R;
      if (s >= FS) continue;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int tap = t + s * NT;
        if (tap < 15) out[(long)(p.co_base + c) * NtotP + ci * 225 + tap * 15 + l16] = acc[t][r];
      }
    }
  }
  if (do_bias) {       // copy 0 (lanes l16 < R) saw every dY row exactly once
    float v = bs;
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (kq == 0 && l16 < R) out[(long)(p.co_base + l16) * NtotP + p.Ntot] = v;
  }
}

struct Wg15Plan {
  int NBC, CIW, COT, coTiles, ciGroups, TH, TW, DP, tilesY, tilesX, IH, IW, DCP, S, TX64, TD64, quad;
  int ga;    // conv_wgrad15g_kernel: dY operand from global memory, LDS holds the X tile only
  // dY-from-global launches: couts [0, 32 n32) in 32-cout tiles, then an optional 16-cout tile, then an optional
  // tap-folded remainder of fold_R <= 8 couts (conv_wgrad15f_kernel) with its own row tiling
  int n32, has16, fold_R, fold_NT, fTH, ftilesY, fIH, fTX64;
  size_t lds_bytes, flds_bytes;
  bool ok;
};

Wg15Plan plan_wgrad15(const mpa_conv_desc* d) {
  Wg15Plan pl{};
  pl.ok = false;
  if (d->kh != 15 || d->kw != 15 || d->sh != 1 || d->sw != 1 || d->ph != 7 || d->pw != 7) return pl;
  const int OH = d->H, OW = d->W;
  pl.NBC = d->Cout <= 16 ? 1 : 2;
  pl.CIW = 1;
  pl.COT = pl.NBC * 16;
  pl.coTiles = (int)mpa_cdiv(d->Cout, pl.COT);
  pl.ciGroups = (int)mpa_cdiv(d->Cin, 4 * pl.CIW);
  const long budget = 78 * 1024;      // two workgroups per CU (160 KiB LDS)
  double bestcost = 1e300;
  // dY-from-global variant: whole-width quads, at most 7 groups of 16 pixels per row (X pitch 128)
  if (OW % 4 == 0 && !getenv("MPA_WG15_LDS_DY")) {
    for (int txn = 1; txn <= 8; ++txn) {
      const int TW = (int)mpa_cdiv(mpa_cdiv(OW, txn), 4) * 4;
      if (TW > 112 || TW < 16 || (long)TW * txn != OW) continue;      // exact tiling: no per-lane column bounds
      for (int TH = std::min(OH, 25); TH >= 1; --TH) {
        const int IH = TH + 14;
        const long tx64 = (long)4 * IH * W15_PITCH;
        if (tx64 * 4 > budget) continue;
        const int ty = (int)mpa_cdiv(OH, TH);
        const double mfma = (double)TH * (TW / 4) * pl.NBC * 15 * 32.0;
        const double stage = (double)tx64 / 64.0 * 80.0 / 4.0;
        const double cost = (double)ty * txn * (mfma + stage + 3000.0);
        if (cost < bestcost) {
          bestcost = cost;
          pl.TH = TH; pl.TW = TW; pl.DP = TW; pl.tilesY = ty; pl.tilesX = txn; pl.IH = IH; pl.IW = W15_PITCH; pl.DCP = 0;
          pl.quad = 1; pl.ga = 1;
          pl.TX64 = (int)tx64; pl.TD64 = 0; pl.lds_bytes = (size_t)tx64 * 4; pl.ok = true;
        }
      }
    }
  }
  for (int txn = 1; txn <= OW && !pl.ga; ++txn) {
    const int TW = (int)mpa_cdiv(OW, txn);
    const int DP = (int)mpa_cdiv(TW, 4) * 4;
    // 16-byte LDS-DMA needs every tile origin and the tensor width 4-aligned; the X window then spans DP+15 columns
    const int quad = (OW % 4 == 0 && TW % 4 == 0) ? 1 : 0;
    const int IW = DP + 14 + quad;
    if (IW > W15_PITCH) continue;
    for (int TH = std::min(OH, 32); TH >= 1; --TH) {
      const int IH = TH + 14;
      const int DCP = quad ? round_mod(TH * DP, 32, 4) : round_mod(TH * DP, 32, 2);
      const long tx64 = mpa_cdiv((long)4 * pl.CIW * IH * W15_PITCH, 64) * 64, td64 = mpa_cdiv((long)pl.COT * DCP, 64) * 64;
      if ((tx64 + td64) * 4 > budget) continue;
      const int ty = (int)mpa_cdiv(OH, TH);
      const double mfma = (double)TH * (DP / 4) * pl.NBC * pl.CIW * 15 * 32.0;
      const double stage = (double)(tx64 + td64) / 64.0 * 80.0 / 4.0;      // LDS-DMA issue cost per wave
      const double cost = (double)ty * txn * (mfma + stage + 3000.0);
      if (cost < bestcost) {
        bestcost = cost;
        pl.TH = TH; pl.TW = TW; pl.DP = DP; pl.tilesY = ty; pl.tilesX = txn; pl.IH = IH; pl.IW = IW; pl.DCP = DCP;
        pl.quad = quad;
        pl.TX64 = (int)tx64; pl.TD64 = (int)td64; pl.lds_bytes = (size_t)(tx64 + td64) * 4; pl.ok = true;
      }
      break;
    }
    if (txn >= 8 && pl.ok) break;
  }
  if (!pl.ok) return pl;
  const long totalTiles = (long)d->B * pl.tilesY * pl.tilesX;
  const long per_cu = std::max<long>(1, std::min<long>(2, (160 * 1024) / (long)pl.lds_bytes));
  // Couts split over up to three launches that share S (the workspace slices): full 32-cout tiles; a 16-cout tile when
  // 9..24 couts are left (16 instead of 32 rows of MFMA work for a half-filled tile); and whatever is then left, at
  // most 8 couts, tap-folded (70 = 2 x 32 + fold 6 for DRCNN:L's prefilters, 20 = 16 + fold 4, 40 = 32 + fold 8).
  pl.n32 = pl.coTiles; pl.has16 = 0; pl.fold_R = 0;
  if (pl.ga && !getenv("MPA_WG15_NOFOLD")) {
    pl.n32 = d->Cout / 32;
    int rem = d->Cout - 32 * pl.n32;
    if (rem > 24) { pl.n32 += 1; rem = 0; }
    if (rem > 8) { pl.has16 = 1; rem = rem > 16 ? rem - 16 : 0; }
    pl.fold_R = rem;
  } else if (pl.ga) {      // the previous rule: a last 32-cout tile filled by at most half runs as a 16-cout tile
    const int rem = d->Cout % 32;
    if (pl.NBC == 2 && rem > 0 && rem <= 16 && pl.coTiles > 1) { pl.n32 = pl.coTiles - 1; pl.has16 = 1; }
    else if (pl.NBC == 1) { pl.n32 = 0; pl.has16 = 1; }
  }
  if (pl.fold_R) {
    // the fold's row tiling: OH + (FS - 1) NT virtual rows, IH = TH + NT - 1 rows of X per tile
    pl.fold_NT = pl.fold_R <= 4 ? 4 : 8;
    const int FS = (15 + pl.fold_NT - 1) / pl.fold_NT, OHv = OH + (FS - 1) * pl.fold_NT;
    double best = 1e300;
    for (int TH = std::min(OHv, 64); TH >= 1; --TH) {
      const int IH = TH + pl.fold_NT - 1;
      const long tx64 = (long)4 * IH * W15_PITCH;
      if (tx64 * 4 > budget) continue;
      const int ty = (int)mpa_cdiv(OHv, TH);
      const double mfma = (double)TH * (pl.TW / 4) * pl.fold_NT * 32.0;
      const double stage = (double)tx64 / 64.0 * 80.0 / 4.0;
      const double cost = (double)ty * (mfma + stage + 3000.0);
      if (cost < best) { best = cost; pl.fTH = TH; pl.ftilesY = ty; pl.fIH = IH; pl.fTX64 = (int)tx64; pl.flds_bytes = (size_t)tx64 * 4; }
    }
  }
  const long slots = 256 * per_cu, groups = (long)pl.ciGroups * std::max(1, pl.n32);
  long S = std::max<long>(1, (2 * slots) / groups);
  if (S > totalTiles) S = totalTiles;
  if (S > 1024) S = 1024;
  // With few tiles per slice the rounding decides: 192 tiles over 128 slices are 2 tiles for half of the workgroups and
  // 1 for the others (measured 107 instead of 125 TFLOP/s for the 32->16 layer at local batch 32).  Among slice counts
  // from half to twice the target, take the cheapest by a small model calibrated on scratch/wg15_s.sh: the busiest CU
  // works through ceil(workgroups / 256) workgroups of ceil(tiles / S) tiles (a tile = its MFMA cycles + 10 % staging;
  // + 8 % when a CU holds a single workgroup and nothing overlaps its staging), and every slice costs one write and one
  // read of its partial sums in the reduction (0.9 us for 128 couts x 16 channels, which is why the large layers want
  // few slices and the 16-cout layers many).
  if (totalTiles / S < 8 && !getenv("MPA_WG15_S_OLD")) {
    const long lo = std::max<long>(1, S / 2), hi = std::min<long>(std::min<long>(2 * S, totalTiles), 1024);
    const double t_tile = 1.1 * (double)pl.TH * (pl.TW / 4) * pl.NBC * 15 * 32.0 / 2.4e9;
    const double t_slice = (double)d->Cout * (d->Cin * 225 + 1) * 8.0 / 4.0e12;
    double best = 1e300;
    long bestS = S;
    for (long c = lo; c <= hi; ++c) {
      const long per_cu_wgs = mpa_cdiv(c * groups, 256);
      const double est = (double)per_cu_wgs * (double)mpa_cdiv(totalTiles, c) * t_tile * (per_cu_wgs < 2 ? 1.08 : 1.0) +
                         (double)c * t_slice;
      if (est < best) { best = est; bestS = c; }
    }
    S = bestS;
  }
  if (const char* e = getenv("MPA_WG15_S")) {      // diagnostics: force the slice count
    const long f = atol(e);
    if (f >= 1 && f <= std::min<long>(totalTiles, 1024)) S = f;
  }
  pl.S = (int)S;
  return pl;
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

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

int64_t mpa_conv2d_packed_floats(const mpa_conv_desc* d, int mode) {
  if (!d) return MPA_ERR_ARG;
  FwdPlan pl;
  int kh = d->kh, kw = d->kw;
  if (mode == 0) {
    pl = plan_fwd(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
  } else {
    BwdDataGeom g = bwd_data_geom(d);
    if (!g.ok) return MPA_ERR_UNSUPPORTED;
    pl = plan_bwd_data(d, g);
    kh = g.kh; kw = g.kw;
  }
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)pl.coTiles * pl.nChunks * kh * (pl.KWS ? pl.KWP : kw) * pl.CK * pl.COTP;
}

static int pack_params(const mpa_conv_desc* d, int mode, const float* w, float* w_packed, PackParams& p);

int mpa_conv2d_pack(const mpa_conv_desc* d, int mode, const float* w, float* w_packed, void* stream) {
  if (!d || !w || !w_packed) return MPA_ERR_ARG;
  PackParams p{};
  const int rc = pack_params(d, mode, w, w_packed, p);
  if (rc) return rc;
  const int blocks = (int)std::min<long>(mpa_cdiv(p.total, 256), 4096);
  MPA_LAUNCH(conv_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  return mpa_launch_status();
}

int mpa_conv2d_pack_entry_bytes(void) { return (int)sizeof(PackParams); }

int mpa_conv2d_pack_entry(const mpa_conv_desc* d, int mode, const float* w, float* w_packed, void* host_entry) {
  if (!d || !w || !w_packed || !host_entry) return MPA_ERR_ARG;
  PackParams p{};
  const int rc = pack_params(d, mode, w, w_packed, p);
  if (rc) return rc;
  memcpy(host_entry, &p, sizeof(PackParams));
  return MPA_OK;
}

int mpa_conv2d_pack_many(const void* device_table, int n, void* stream) {
  if (!device_table || n < 0) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  if (n > 65535) return MPA_ERR_ARG;
  MPA_LAUNCH(conv_pack_many_kernel, dim3(64, (unsigned)n), dim3(256), 0, (hipStream_t)stream, (const PackParams*)device_table);
  return mpa_launch_status();
}

static int pack_params(const mpa_conv_desc* d, int mode, const float* w, float* w_packed, PackParams& p) {
  p.w = w; p.wp = w_packed;
  p.Cout_w = d->Cout; p.Cin_w = d->Cin; p.kh_w = d->kh; p.kw_w = d->kw;
  p.mode = mode; p.xphase = 0;
  FwdPlan pl;
  if (mode == 0) {
    pl = plan_fwd(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
    p.CinP = d->Cin; p.CoutP = d->Cout; p.kh = d->kh; p.kw = d->kw;
  } else {
    BwdDataGeom g = bwd_data_geom(d);
    if (!g.ok) return MPA_ERR_UNSUPPORTED;
    pl = plan_bwd_data(d, g);
    p.CinP = g.Cin; p.CoutP = g.Cout; p.kh = g.kh; p.kw = g.kw; p.xphase = g.xphase ? 1 : 0; p.yphase = g.yphase;
  }
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  p.CK = pl.CK; p.nChunks = pl.nChunks; p.COT = pl.COT; p.COTP = pl.COTP; p.coTiles = pl.coTiles;
  p.KWP = pl.KWS ? pl.KWP : 0;
  p.total = (long)pl.coTiles * pl.nChunks * p.kh * (pl.KWS ? pl.KWP : p.kw) * pl.CK * pl.COTP;
  return MPA_OK;
}

static int conv_fwd_impl(int B, int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw,
                         const float* x, const float* wp, const float* bias, float* y, int act, float slope,
                         long outBS, long outCS, int outRS, int outXmul, int outCdiv, hipStream_t s, bool allow_split = false,
                         int Hplan = 0, int outYmul = 1, int outH = 0, float* stats = nullptr) {
  FwdPlan pl = plan_fwd(B, Cin, Hplan ? Hplan : H, W, Cout, kh, kw, sh, sw, ph, pw, allow_split, outCdiv < Cout,
                        outCdiv < Cout ? outXmul : 1);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  if (stats) {
    if (pl.KS > 1 || act != MPA_ACT_NONE) return MPA_ERR_UNSUPPORTED;
    pl.lds_bytes = std::max<size_t>(pl.lds_bytes, 8192);      // the epilogue's reduction scratch (tiny tiles stage less)
  }
  ConvFwdParams p{};
  p.x = x; p.wp = wp; p.bias = bias; p.y = y;
  p.B = B; p.Cin = Cin; p.H = H; p.W = W; p.Cout = Cout; p.OH = pl.OH; p.OW = pl.OW;
  p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw;
  p.TH = pl.TH; p.TW = pl.TW; p.tilesY = pl.tilesY; p.tilesX = pl.tilesX; p.CK = pl.CK; p.nChunks = pl.nChunks;
  p.IH = pl.IH; p.IW = pl.IW; p.LW = pl.LW; p.CHP = pl.CHP; p.COT = pl.COT; p.COTP = pl.COTP;
  p.IN64 = (int)(mpa_cdiv((long)pl.CK * pl.CHP, 64) * 64);
  p.SL64 = (int)(mpa_cdiv((long)(pl.KWS ? pl.KWP : kw) * pl.CK * pl.COTP, 64) * 64);
  p.quad = pl.quad;
  { const char* e = getenv("MPA_DEBUG_FWD"); p.dbg = e ? atoi(e) : 0; }
  p.act = act; p.slope = slope;
  p.outBS = outBS; p.outCS = outCS; p.outRS = outRS; p.outXmul = outXmul; p.outCdiv = outCdiv;
  p.outYmul = outYmul; p.outH = outH ? outH : pl.OH;
  p.chunksPer = (int)mpa_cdiv(pl.nChunks, pl.KS);
  p.coTiles = pl.coTiles; p.nTilesAll = B * pl.tilesY * pl.tilesX;
  p.stats = stats;
  if (mpa_cdiv(pl.nChunks, p.chunksPer) <= 1) return launch_fwd(pl, p, s);
  // channel-split launch: slices add into a zeroed output, the activation (if any) runs afterwards in place
  if (mpa_zero_async(y, sizeof(float) * (size_t)B * (size_t)outBS, s) != MPA_OK) return MPA_ERR_LAUNCH;
  p.act = MPA_ACT_NONE;
  const int rc = launch_fwd(pl, p, s);
  if (rc != MPA_OK || act == MPA_ACT_NONE) return rc;
  return mpa_act_fwd(y, y, (int64_t)B * outBS, act, slope, s);
}

int mpa_conv2d_fwd(const mpa_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                   int act, float slope, void* stream) {
  if (!d || !x || !w_packed || !y || d->B <= 0) return MPA_ERR_ARG;
  const int OH = (d->H + 2 * d->ph - d->kh) / d->sh + 1, OW = (d->W + 2 * d->pw - d->kw) / d->sw + 1;
  if (OH <= 0 || OW <= 0) return MPA_ERR_ARG;
  return conv_fwd_impl(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw, x, w_packed, bias,
                       y, act, slope, (long)d->Cout * OH * OW, (long)OH * OW, OW, 1, d->Cout, (hipStream_t)stream);
}

int64_t mpa_conv2d_fwd_stats_rows(const mpa_conv_desc* d) {
  if (!d) return MPA_ERR_ARG;
  FwdPlan pl = plan_fwd(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)d->B * pl.tilesY * pl.tilesX;
}

int mpa_conv2d_fwd_stats(const mpa_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                         float* partials, void* stream) {
  if (!d || !x || !w_packed || !y || !partials || d->B <= 0) return MPA_ERR_ARG;
  const int OH = (d->H + 2 * d->ph - d->kh) / d->sh + 1, OW = (d->W + 2 * d->pw - d->kw) / d->sw + 1;
  if (OH <= 0 || OW <= 0) return MPA_ERR_ARG;
  return conv_fwd_impl(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw, x, w_packed, bias,
                       y, MPA_ACT_NONE, 0.f, (long)d->Cout * OH * OW, (long)OH * OW, OW, 1, d->Cout, (hipStream_t)stream,
                       false, 0, 1, 0, partials);
}

int mpa_conv2d_bwd_data(const mpa_conv_desc* d, const float* dy, const float* w_packed, float* dx, void* stream) {
  if (!d || !dy || !w_packed || !dx || d->B <= 0) return MPA_ERR_ARG;
  BwdDataGeom g = bwd_data_geom(d);
  if (!g.ok) return MPA_ERR_UNSUPPORTED;
  const long inBS = (long)d->Cin * d->H * d->W, inCS = (long)d->H * d->W;
  if (g.yphase > 1) {
    // V output rows per cout block: derived conv with vertical stride V; row oy*V + v of channel cout''/V
    return conv_fwd_impl(d->B, g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, g.sh, 1, g.ph, g.pw, dy, w_packed, nullptr, dx,
                         MPA_ACT_NONE, 0.f, inBS, inCS, d->W, 1, d->Cin, (hipStream_t)stream, true, g.Hplan, g.yphase,
                         d->H);
  }
  if (!g.xphase) {
    // output of the derived conv has size H x W again
    return conv_fwd_impl(d->B, g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, 1, 1, g.ph, g.pw, dy, w_packed, nullptr, dx,
                         MPA_ACT_NONE, 0.f, inBS, inCS, d->W, 1, g.Cout, (hipStream_t)stream, true);
  }
  return conv_fwd_impl(d->B, g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, 1, 1, g.ph, g.pw, dy, w_packed, nullptr, dx,
                       MPA_ACT_NONE, 0.f, inBS, inCS, d->W, d->sw, d->Cin, (hipStream_t)stream, true);
}

int mpa_conv2d_describe_plan(const mpa_conv_desc* d, int mode, char* buf, int buflen) {
  if (!d || !buf || buflen <= 0) return MPA_ERR_ARG;
  if (mode == 2) {
    Wg15Plan q = plan_wgrad15(d);
    if (q.ok) {
      int n = snprintf(buf, buflen, "wgrad15%s<%d,%d> COT=%d coTiles=%d ciGroups=%d tile=%dx%d (DP %d) tiles=%dx%d S=%d lds=%zuB",
                       q.ga ? "g" : "", q.NBC, q.CIW, q.COT, q.coTiles, q.ciGroups, q.TH, q.TW, q.DP, q.tilesY, q.tilesX, q.S, q.lds_bytes);
      if (q.ga && n > 0 && n < buflen)
        snprintf(buf + n, buflen - n, " launches: %dx32 %s fold=%d (NT %d, tile rows %d x %d)", q.n32, q.has16 ? "+16" : "",
                 q.fold_R, q.fold_NT, q.fTH, q.ftilesY);
      return MPA_OK;
    }
    WgPlan w = plan_wgrad(d);
    if (!w.ok) return MPA_ERR_UNSUPPORTED;
    snprintf(buf, buflen, "wgrad%s<%d,%d> COT=%d coTiles=%d nPerBlock=%d nTiles=%d XCH=%d tile=%dx%d (DP %d) tiles=%dx%d S=%d quad=%d ef=%d lds=%zuB",
             w.ga ? "_g" : "", w.NBC, w.NTW, w.COT, w.coTiles, w.nPerBlock, w.nTiles, w.XCH, w.TH, w.TW, w.DP, w.tilesY,
             w.tilesX, w.S, w.quad, w.ef, w.lds_bytes);
    return MPA_OK;
  }
  FwdPlan f;
  if (mode == 0) f = plan_fwd(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
  else {
    BwdDataGeom g = bwd_data_geom(d);
    if (!g.ok) return MPA_ERR_UNSUPPORTED;
    f = plan_bwd_data(d, g);
  }
  if (!f.ok) return MPA_ERR_UNSUPPORTED;
  snprintf(buf, buflen, "fwd<%d,%d> COT=%d coTiles=%d CK=%d chunks=%d tile=%dx%d tiles=%dx%d halo=%dx%d LW=%d quad=%d kwvec=%d ksplit=%d lds=%zuB",
           f.NB, f.PB, f.COT, f.coTiles, f.CK, f.nChunks, f.TH, f.TW, f.tilesY, f.tilesX, f.IH, f.IW, f.LW, f.quad, f.KWS, f.KS, f.lds_bytes);
  return MPA_OK;
}

int64_t mpa_conv2d_bwd_weight_workspace(const mpa_conv_desc* d) {
  if (!d) return MPA_ERR_ARG;
  Wg15Plan p15 = plan_wgrad15(d);
  if (p15.ok) return (int64_t)p15.S * d->Cout * (d->Cin * 225 + 1) * 4;
  WgPlan pl = plan_wgrad(d);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)pl.S * d->Cout * (pl.Ntot + 1) * 4;
}

int mpa_conv2d_bwd_weight(const mpa_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                          void* workspace, int64_t workspace_bytes, void* stream) {
  if (!d || !x || !dy || !dw || d->B <= 0) return MPA_ERR_ARG;
  Wg15Plan p15 = plan_wgrad15(d);
  if (p15.ok) {
    const int Ntot = d->Cin * 225;
    const int64_t need15 = (int64_t)p15.S * d->Cout * (Ntot + 1) * 4;
    if (!workspace || workspace_bytes < need15) return MPA_ERR_WORKSPACE;
    Wg15Params q{};
    q.x = x; q.dy = dy; q.ws = (float*)workspace;
    q.B = d->B; q.Cin = d->Cin; q.H = d->H; q.W = d->W; q.Cout = d->Cout; q.OH = d->H; q.OW = d->W;
    q.COT = p15.COT; q.TH = p15.TH; q.TW = p15.TW; q.DP = p15.DP; q.tilesY = p15.tilesY; q.tilesX = p15.tilesX;
    q.IH = p15.IH; q.IW = p15.IW; q.DCP = p15.DCP; q.S = p15.S; q.Ntot = Ntot; q.TX64 = p15.TX64; q.TD64 = p15.TD64; q.quad = p15.quad;
    { const char* e = getenv("MPA_DEBUG_WG15"); q.dbg = e ? atoi(e) : 0; }
    hipStream_t s15 = (hipStream_t)stream;
    dim3 grid15((unsigned)p15.S, (unsigned)p15.ciGroups, (unsigned)p15.coTiles);
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad15_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      (void)hipFuncSetAttribute((const void*)conv_wgrad15_kernel<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      attr_set = true;
    }
    if (p15.ga) {
      static bool attr_g = false;
      if (!attr_g) {
#define MPA_WG15G_ATTR(...) (void)hipFuncSetAttribute((const void*)conv_wgrad15g_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
        MPA_WG15G_ATTR(1, false); MPA_WG15G_ATTR(2, false); MPA_WG15G_ATTR(1, true); MPA_WG15G_ATTR(2, true);
        MPA_WG15G_ATTR(1, false, true); MPA_WG15G_ATTR(2, false, true); MPA_WG15G_ATTR(1, true, true); MPA_WG15G_ATTR(2, true, true);
#undef MPA_WG15G_ATTR
#define MPA_WG15F_ATTR(...) (void)hipFuncSetAttribute((const void*)conv_wgrad15f_kernel<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024)
        MPA_WG15F_ATTR(8, false, false); MPA_WG15F_ATTR(8, false, true); MPA_WG15F_ATTR(8, true, false); MPA_WG15F_ATTR(8, true, true);
        MPA_WG15F_ATTR(4, false, false); MPA_WG15F_ATTR(4, false, true); MPA_WG15F_ATTR(4, true, false); MPA_WG15F_ATTR(4, true, true);
#undef MPA_WG15F_ATTR
        attr_g = true;
      }
      const bool tl = (p15.DP & 15) != 0;
      const bool ev = ((p15.DP >> 4) & 1) == 0;
#define MPA_WG15G_GO(...) MPA_LAUNCH((conv_wgrad15g_kernel<__VA_ARGS__>), grid15, dim3(256), p15.lds_bytes, s15, q)
#define MPA_WG15F_GO(...) MPA_LAUNCH((conv_wgrad15f_kernel<__VA_ARGS__>), grid15, dim3(256), p15.flds_bytes, s15, q)
      if (p15.n32) {
        grid15.z = (unsigned)p15.n32; q.co_base = 0; q.COT = 32;
        if (tl) { if (ev) MPA_WG15G_GO(2, true, true); else MPA_WG15G_GO(2, true, false); }
        else { if (ev) MPA_WG15G_GO(2, false, true); else MPA_WG15G_GO(2, false, false); }
      }
      if (p15.has16) {
        grid15.z = 1; q.co_base = 32 * p15.n32; q.COT = 16;
        if (tl) { if (ev) MPA_WG15G_GO(1, true, true); else MPA_WG15G_GO(1, true, false); }
        else { if (ev) MPA_WG15G_GO(1, false, true); else MPA_WG15G_GO(1, false, false); }
      }
      if (p15.fold_R) {
        grid15.z = 1; q.co_base = 32 * p15.n32 + 16 * p15.has16; q.COT = 16; q.fold_R = p15.fold_R;
        q.TH = p15.fTH; q.tilesY = p15.ftilesY; q.IH = p15.fIH; q.TX64 = p15.fTX64;
        if (p15.fold_NT == 8) {
          if (tl) { if (ev) MPA_WG15F_GO(8, true, true); else MPA_WG15F_GO(8, true, false); }
          else { if (ev) MPA_WG15F_GO(8, false, true); else MPA_WG15F_GO(8, false, false); }
        } else {
          if (tl) { if (ev) MPA_WG15F_GO(4, true, true); else MPA_WG15F_GO(4, true, false); }
          else { if (ev) MPA_WG15F_GO(4, false, true); else MPA_WG15F_GO(4, false, false); }
        }
      }
#undef MPA_WG15F_GO
#undef MPA_WG15G_GO
    } else
    if (p15.NBC == 1) MPA_LAUNCH((conv_wgrad15_kernel<1, 1>), grid15, dim3(256), p15.lds_bytes, s15, q);
    else MPA_LAUNCH((conv_wgrad15_kernel<2, 1>), grid15, dim3(256), p15.lds_bytes, s15, q);
    int rc15 = mpa_launch_status();
    if (rc15) return rc15;
    const long n15 = (long)d->Cout * (Ntot + 1);
    MPA_LAUNCH(reduce_partials_kernel, dim3((unsigned)std::min<long>(mpa_cdiv(n15, 256), 2048)), dim3(256), 0, s15,
               (const float*)workspace, dw, db, d->Cout, Ntot, Ntot + 1, p15.S);
    return mpa_launch_status();
  }
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
  hipStream_t s = (hipStream_t)stream;
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
  const long n = (long)d->Cout * (pl.Ntot + 1);
  MPA_LAUNCH(reduce_partials_kernel, dim3((unsigned)std::min<long>(mpa_cdiv(n, 256), 2048)), dim3(256), 0, s,
             (const float*)workspace, dw, db, d->Cout, pl.Ntot, pl.Ntot + 1, pl.S);
  return mpa_launch_status();
}

}  // extern "C"
