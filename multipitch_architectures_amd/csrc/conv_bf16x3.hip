// Split-bf16 ("bf16x3") convolution for gfx950: every fp32 operand is carried as hi + lo bf16 halves and a product is
// hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- 3 MFMAs at 16x the fp32-input MFMA rate,
// error ~2e-5 of the result's rms on this model's data (scratch/split_bf16_error.py), inside the 1e-4 forward bound.
// Opt-in (ops.set_conv_precision("bf16x3")); the exact-fp32 path of conv.hip stays the default and the headline.
//
// Replaces the same call sites as conv.hip for the 15-row filters (all three passes) and the 9-row filters (forward and
// backward-data): double_conv (unet_cnns.py:49-59) of the 75x216, 37x108 and 18x54 levels, conv1 / prefilt_list
// (basic_cnns.py:371-387).
//
// Layouts
//   split activations  xs[b][c/8][hi|lo][y][x][8 channels]  bf16 -- one 16-byte granule = 8 channels of one pixel, the
//                      K fragment one lane feeds to the MFMA; written by split_bf16_kernel (or a producing epilogue)
//   packed filters     wp[cout tile 16][chunk 8 ch][q = dx quad][dy][hi|lo][lane 64][8 ch] -- one 1 KiB block is the A
//                      operand of one MFMA exactly as the 64 lanes read it
// Kernel (forward and, with flipped/transposed filters, backward-data): D[cout 16][pixel 16] per MFMA, K = 32 =
// 8 channels x 4 horizontal taps.  A wave owns R output rows x 16 columns for one 16-cout tile and *slides* over the
// input rows: the filter fragments of all KH vertical taps of the current (chunk, dx quad) sit in registers
// (KH x 2 x 4 VGPRs), and input row d, read from LDS once, feeds every (output row, dy) pair with row + dy = d --
// up to min(R, KH) x 3 MFMAs per pair of 16-byte LDS reads, which is what makes a 16-cout tile affordable
// (without the slide a B fragment feeds 3 MFMAs and the LDS pipe, not the matrix pipe, sets the speed).
// Bank conflicts: lanes n -> pixel f(n) (even pixels for n in {0-3,12-15}, odd for {4-11}) and k-groups g -> taps
// (0,2,1,3) make every 16-lane ds_read_b128 group touch 16 distinct 16-byte slots.
#include "mpa_common.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __attribute__((aligned(16))) uint4 bfx_zero[64];      // source of every out-of-image granule

__device__ __forceinline__ void glds16(const uint4* src, uint4* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------ operand split
// x fp32 [B][C][HW] -> out [B][C8][2][HW] granules of 8 bf16 (channels 8*c8 .. 8*c8+7 of one pixel; hi plane, lo plane)
__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, uint4* __restrict__ out, int C, int C8,
                                                         long HW, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long pix = i % HW, t = i / HW;
    const int c8 = (int)(t % C8);
    const long b = t / C8;
    const float* src = x + (b * C + (long)c8 * 8) * HW + pix;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {       // clamped address + mask: all eight loads go out together (no branch per element)
      const int cj = c8 * 8 + j;
      const float t0 = src[(long)(cj < C ? j : 0) * HW];
      v[j] = cj < C ? t0 : 0.f;
    }
    bf16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      hi[j] = (__bf16)v[j];
      lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
    uint4* o = out + ((b * C8 + c8) * 2) * HW + pix;
    o[0] = __builtin_bit_cast(uint4, hi);
    o[HW] = __builtin_bit_cast(uint4, lo);
  }
}

// ------------------------------------------------------------------------------------------------ filter packing
struct BfxPackParams {
  const float* w;
  uint4* wp;
  int Cout_w, Cin_w, kh, kw;     // original filter (Cout_w, Cin_w, kh, kw)
  int mode;                      // 0: forward, 1: backward-data (flipped taps, channels transposed)
  int CoutP, CinP;               // dims of the convolution that consumes the bank
  int coTiles, nChunks, QN;
  long total;                    // granules
};

__device__ __forceinline__ int bfx_tap_of_group(int g) { return g == 0 ? 0 : (g == 1 ? 2 : (g == 2 ? 1 : 3)); }

__global__ __launch_bounds__(256) void bfx_pack_kernel(const BfxPackParams p) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < p.total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int lane = (int)(r & 63); r >>= 6;
    const int hl = (int)(r & 1); r >>= 1;
    const int dy = (int)(r % p.kh); r /= p.kh;
    const int q = (int)(r % p.QN); r /= p.QN;
    const int chunk = (int)(r % p.nChunks); r /= p.nChunks;
    const int cot = (int)r;
    const int m = lane & 15, g = lane >> 4;
    const int co = cot * 16 + m, dx = 4 * q + bfx_tap_of_group(g);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = chunk * 8 + j;
      float v = 0.f;
      if (co < p.CoutP && ci < p.CinP && dx < p.kw) {
        if (p.mode == 0) v = p.w[(((long)co * p.Cin_w + ci) * p.kh + dy) * p.kw + dx];
        else v = p.w[(((long)ci * p.Cin_w + co) * p.kh + (p.kh - 1 - dy)) * p.kw + (p.kw - 1 - dx)];
      }
      const __bf16 h = (__bf16)v;
      o[j] = hl ? (__bf16)(v - (float)h) : h;
    }
    p.wp[i] = __builtin_bit_cast(uint4, o);
  }
}

// ------------------------------------------------------------------------------------------------ convolution
constexpr int BFX_PX = 128;        // LDS row pitch in granules: 7 waves x 16 columns + 15 taps, rounded up
constexpr int BFX_MAXW = 7;        // waves (column blocks) per workgroup

struct BfxParams {
  const uint4* xs;
  const uint4* wp;
  const float* bias;
  float* y;
  float* stats;              // nullable: [nTilesAll][Cout][2] partial sums of y and y^2 (BatchNorm fusion)
  int B, C8, H, W, Cout, OH, OW, QN, ph, pw;
  int tilesY, tilesX, coTiles, nTilesAll, NW, act;
  float slope;
  int dbg;                   // diagnostics (env MPA_BFX_DEBUG): 1 = stage the input tile once, 2 = + the filter slab once,
                             // 3 = + skip the MFMA sweep (staging / synchronisation skeleton only); results are wrong
};

template <int KH, int R>
__global__ __launch_bounds__(512) void conv_bfx_kernel(const BfxParams p) {
  constexpr int XR = R + KH - 1;                  // input rows of a tile
  constexpr int XG = XR * 2 * BFX_PX;             // granules of the X image [row][hi|lo][BFX_PX]
  constexpr int AG = KH * 2 * 64;                 // granules of one filter slab [dy][hi|lo][lane]
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  uint4* lds_x = lds;
  uint4* lds_a = lds + XG;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // XCD-aware order (as conv_fwd_kernel): the coTiles workgroups that stage the same input tile run back to back on one XCD
  const int w = blockIdx.x;
  const int seq = w >> 3;
  const int cot = seq % p.coTiles;
  int bid = (seq / p.coTiles) * 8 + (w & 7);
  if (bid >= p.nTilesAll) return;
  const int ptile = bid;
  const int tx = bid % p.tilesX;
  bid /= p.tilesX;
  const int ty = bid % p.tilesY;
  const int b = bid / p.tilesY;
  const int oy0 = ty * R, ox0 = tx * (16 * p.NW);
  const int iy0 = oy0 - p.ph, ix0 = ox0 - p.pw;
  const int n = lane & 15, g = lane >> 4;
  const int fn = n < 4 ? 2 * n : (n >= 12 ? 2 * (n - 8) : 2 * (n - 4) + 1);     // pixel of MFMA column n
  const int xlane = wave * 16 + fn + bfx_tap_of_group(g);

  f32x4 acc[R];
#pragma unroll
  for (int i = 0; i < R; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long plane = (long)p.H * p.W;
  const uint4* xsb = p.xs + (long)b * p.C8 * 2 * plane;
  const int nslab = p.C8 * p.QN;
  const uint4* wtile = p.wp + (long)cot * nslab * AG;
  const uint4* zsrc = bfx_zero + (lane & 15);

  auto stage_x = [&](int chunk) {
    const uint4* src0 = xsb + (long)chunk * 2 * plane;
    for (int i = wave; i < XG / 64; i += p.NW) {
      const int e = i * 64 + lane;
      const int px = e & (BFX_PX - 1), t = e >> 7;
      const int hl = t & 1, d = t >> 1;
      const int gy = iy0 + d, gx = ix0 + px;
      const bool ok = (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
      const long off = (long)hl * plane + (long)gy * p.W + gx;
      glds16(ok ? src0 + off : zsrc, lds_x + i * 64);
    }
  };
  auto stage_a = [&](int s) {
    const uint4* src = wtile + (long)s * AG + lane;
    for (int i = wave; i < KH * 2; i += p.NW) glds16(src + i * 64, lds_a + i * 64);
  };

  stage_x(0);
  stage_a(0);
  int s = 0;
  for (int chunk = 0; chunk < p.C8; ++chunk) {
    for (int q = 0; q < p.QN; ++q, ++s) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // slab s (and, at q == 0, the chunk's input tile) has landed
      __builtin_amdgcn_s_barrier();
      bf16x8 ah[KH], al[KH];
#pragma unroll
      for (int dy = 0; dy < KH; ++dy) {
        ah[dy] = __builtin_bit_cast(bf16x8, lds_a[(dy * 2 + 0) * 64 + lane]);
        al[dy] = __builtin_bit_cast(bf16x8, lds_a[(dy * 2 + 1) * 64 + lane]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                         // every wave holds the slab in registers: its buffer is free
      if (s + 1 < nslab && MPA_DBG(p) < 2) stage_a(s + 1);       // ... and the next slab streams in behind the MFMAs
      const uint4* xb = lds_x + xlane + 4 * q;
      if (MPA_DBG(p) == 3) continue;
      // input row d+1 is requested before the MFMAs of row d are issued: its LDS latency hides behind them
      bf16x8 bh = __builtin_bit_cast(bf16x8, xb[0]);
      bf16x8 bl = __builtin_bit_cast(bf16x8, xb[BFX_PX]);
#pragma unroll
      for (int d = 0; d < XR; ++d) {
        bf16x8 nh = bh, nl = bl;
        if (d + 1 < XR) {
          nh = __builtin_bit_cast(bf16x8, xb[(d * 2 + 2) * BFX_PX]);
          nl = __builtin_bit_cast(bf16x8, xb[(d * 2 + 3) * BFX_PX]);
        }
#pragma unroll
        for (int pb = 0; pb < R; ++pb) {
          const int dy = d - pb;
          if (dy >= 0 && dy < KH) {
            acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[dy], bh, acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[dy], bl, acc[pb], 0, 0, 0);
            acc[pb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[dy], bh, acc[pb], 0, 0, 0);
          }
        }
        bh = nh; bl = nl;
      }
    }
    if (chunk + 1 < p.C8) {
      __builtin_amdgcn_s_barrier();                         // every wave is done with this chunk's input tile
      if (MPA_DBG(p) < 1) stage_x(chunk + 1);
    }
  }

  // epilogue: lane (n, g) holds couts 4g .. 4g+3 of pixel (row pb, column fn) of each of its R rows
  const int ox = ox0 + wave * 16 + fn;
  const bool colok = ox < p.OW;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, qsum[4] = {0.f, 0.f, 0.f, 0.f};
  float bs[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int co = cot * 16 + 4 * g + r;
    bs[r] = (p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f;
  }
  float* yb = p.y + (long)b * p.Cout * p.OH * p.OW;
#pragma unroll
  for (int pb = 0; pb < R; ++pb) {
    const int oy = oy0 + pb;
    const bool ok = colok && oy < p.OH;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cot * 16 + 4 * g + r;
      float v = acc[pb][r] + bs[r];
      if (ok) { ssum[r] += v; qsum[r] += v * v; }
      v = mpa_apply_act(v, p.act, p.slope);
      if (ok && co < p.Cout) yb[((long)co * p.OH + oy) * p.OW + ox] = v;
    }
  }
  if (p.stats) {
    __syncthreads();                                        // main-loop LDS images are dead
    float* red = reinterpret_cast<float*>(lds);             // [wave][16 couts][2]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float sv = ssum[r], qv = qsum[r];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { sv += __shfl_xor(sv, o, 64); qv += __shfl_xor(qv, o, 64); }
      if (n == 0) { red[(wave * 16 + 4 * g + r) * 2] = sv; red[(wave * 16 + 4 * g + r) * 2 + 1] = qv; }
    }
    __syncthreads();
    const int co = cot * 16 + (int)threadIdx.x;
    if (threadIdx.x < 16 && co < p.Cout) {
      float s4 = 0.f, q4 = 0.f;
      for (int wv = 0; wv < p.NW; ++wv) { s4 += red[(wv * 16 + threadIdx.x) * 2]; q4 += red[(wv * 16 + threadIdx.x) * 2 + 1]; }
      *reinterpret_cast<float2*>(p.stats + ((long)ptile * p.Cout + co) * 2) = make_float2(s4, q4);
    }
  }
}

struct BfxPlan {
  bool ok;
  int R, NW, tilesY, tilesX, coTiles, C8, QN;
  size_t lds_bytes;
};

// the convolution described by (Cin, H, W, Cout, kh, kw, ph, pw), stride 1
BfxPlan bfx_plan(int Cin, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw) {
  BfxPlan pl{};
  pl.ok = false;
  if ((kh != 15 && kh != 9) || kw < 1 || kw > 16 || sh != 1 || sw != 1) return pl;
  const int OH = H + 2 * ph - kh + 1, OW = W + 2 * pw - kw + 1;
  if (OH <= 0 || OW <= 0) return pl;
  const int forceR = mpa_diag().bfx_r;      // diagnostics
  // rows per wave: the candidate with the least padded rows (ties: the taller one, fewer filter-fragment reloads)
  const int cand15[2] = {15, 13}, cand9[3] = {19, 18, 13};
  const int* cand = kh == 15 ? cand15 : cand9;
  const int ncand = kh == 15 ? 2 : 3;
  long bestpad = 1L << 60;
  for (int i = 0; i < ncand; ++i) {
    const long pad = mpa_cdiv(OH, cand[i]) * cand[i];
    if (pad < bestpad) { bestpad = pad; pl.R = cand[i]; }
  }
  for (int i = 0; i < ncand; ++i)
    if (forceR == cand[i]) pl.R = forceR;
  pl.tilesY = (int)mpa_cdiv(OH, pl.R);
  const int cb = (int)mpa_cdiv(OW, 16);
  pl.tilesX = (int)mpa_cdiv(cb, BFX_MAXW);
  pl.NW = (int)mpa_cdiv(cb, pl.tilesX);
  pl.coTiles = (int)mpa_cdiv(Cout, 16);
  pl.C8 = (int)mpa_cdiv(Cin, 8);
  pl.QN = (int)mpa_cdiv(kw, 4);
  pl.lds_bytes = ((size_t)(pl.R + kh - 1) * 2 * BFX_PX + (size_t)kh * 2 * 64) * 16;
  pl.ok = true;
  return pl;
}

struct BfxGeom { bool ok; int Cin, H, W, Cout, kh, kw, ph, pw; };

BfxGeom bfx_geom(const mpa_conv_desc* d, int mode) {
  BfxGeom g{};
  g.ok = false;
  if (!d || d->sh != 1 || d->sw != 1) return g;
  const int OH = d->H + 2 * d->ph - d->kh + 1, OW = d->W + 2 * d->pw - d->kw + 1;
  if (OH <= 0 || OW <= 0) return g;
  if (mode == 0) {
    g = BfxGeom{true, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->ph, d->pw};
  } else {
    g = BfxGeom{true, d->Cout, OH, OW, d->Cin, d->kh, d->kw, d->kh - 1 - d->ph, d->kw - 1 - d->pw};
    if (g.ph < 0 || g.pw < 0) g.ok = false;
  }
  return g;
}

int bfx_launch(const BfxGeom& g, int B, const void* xs, const void* wp, const float* bias, float* y, int act, float slope,
               float* stats, hipStream_t s) {
  const BfxPlan pl = bfx_plan(g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, 1, 1, g.ph, g.pw);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  BfxParams p{};
  p.xs = (const uint4*)xs; p.wp = (const uint4*)wp; p.bias = bias; p.y = y; p.stats = stats;
  p.B = B; p.C8 = pl.C8; p.H = g.H; p.W = g.W; p.Cout = g.Cout;
  p.OH = g.H + 2 * g.ph - g.kh + 1; p.OW = g.W + 2 * g.pw - g.kw + 1;
  p.QN = pl.QN; p.ph = g.ph; p.pw = g.pw;
  p.tilesY = pl.tilesY; p.tilesX = pl.tilesX; p.coTiles = pl.coTiles; p.nTilesAll = B * pl.tilesY * pl.tilesX;
  p.NW = pl.NW; p.act = act; p.slope = slope;
  p.dbg = mpa_diag().dbg_bfx;
  const dim3 grid((unsigned)(mpa_cdiv(p.nTilesAll, 8) * 8 * pl.coTiles));
  static bool attr = false;
  if (!attr) {
#define BFX_ATTR(KH_, R_) (void)hipFuncSetAttribute((const void*)conv_bfx_kernel<KH_, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)
    BFX_ATTR(15, 15); BFX_ATTR(15, 13); BFX_ATTR(9, 19); BFX_ATTR(9, 18); BFX_ATTR(9, 13);
#undef BFX_ATTR
    attr = true;
  }
#define BFX_GO(KH_, R_) MPA_LAUNCH((conv_bfx_kernel<KH_, R_>), grid, dim3(64 * pl.NW), pl.lds_bytes, s, p)
  if (g.kh == 15) { if (pl.R == 15) BFX_GO(15, 15); else BFX_GO(15, 13); }
  else { if (pl.R == 19) BFX_GO(9, 19); else if (pl.R == 18) BFX_GO(9, 18); else BFX_GO(9, 13); }
#undef BFX_GO
  return mpa_launch_status();
}


// ------------------------------------------------------------------------------------------------ backward-weight
// dW[co][ci][dy][dx] = sum over pixels of dY[co][pixel] * X[ci][pixel + tap]: D[co 16][ci 16] per MFMA and tap, K = 32
// consecutive pixels of one row.  Both operands want "lane = channel, 8 consecutive pixels" while the split tensors hold
// "granule = pixel, 8 consecutive channels": ds_read_b64_tr_b16 transposes on the way out of LDS (two reads per fragment).
// A workgroup = one (16-cout block, 16-channel block) and a slice of the column strips (image b, 32 columns); it walks a
// strip top to bottom, WG_R rows of dY per step, with the input rows in a 32-row LDS ring (rows y-7 .. y+WG_R+6 live, the
// next WG_R streaming in behind the MFMAs).  Wave v owns the horizontal taps dx = v and v + 8 and all 15 vertical ones
// (2 x 15 accumulators): input row d, read once per dx, feeds every (dY row, dy) pair with row + dy = d -- the same slide
// as the forward kernel.  The 16th dx slot (wave 7) accumulates the bias gradient (dY times ones) instead.
// Partial sums per slice go to the workspace [S][Cout][Cin*225 + 1] (every element written exactly once), summed in a
// fixed order by bfx_reduce_partials_kernel: run-to-run reproducible, no atomics.
constexpr int WG_R = 8;
constexpr int WG_RING = 32;                 // ring rows: slot(row) = (row + 9) & 31, so that 8-row blocks never wrap
constexpr int WG_XP = 52;                   // X line pitch in granules: 32 + 14 columns, == 4 (mod 16): the two 8-channel
constexpr int WG_DP = 36;                   // planes of a pixel sit 64 bytes apart (mod 256) -> conflict-free tr reads
constexpr int WG_XROW = 4 * WG_XP;          // granules of one ring row: [hi|lo][c8 0|1][WG_XP]
constexpr int WG_DROW = 4 * WG_DP;
constexpr int WG_XG = WG_RING * WG_XROW;    // 6656 granules
constexpr int WG_DG = WG_R * WG_DROW;       // 1152 granules per dY buffer (two buffers)

struct BfxWgParams {
  const uint4* xs;        // split input  [B][XC8][2][H][W]
  const uint4* dys;       // split grad   [B][DC8][2][OH][OW]
  float* ws;              // [S][Cout][NtotP]
  int B, Cin, XC8, H, W, Cout, DC8, OH, OW, kw, ph, pw;
  int tilesX, strips, S, steps, NtotP, ciBlocks;
};

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bf16x8 bfx_tr_frag(const char* lds_addr) {
  // two transposing reads: pixels 8g .. 8g+3 and 8g+4 .. 8g+7 of this lane's channel
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_addr));
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(lds_addr + 64));
  bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}

template <int KH>
__global__ __launch_bounds__(512) void conv_bfx_wgrad_kernel(const BfxWgParams p) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  uint4* lds_x = lds;
  uint4* lds_d = lds + WG_XG;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int slice = blockIdx.x, cib = blockIdx.y, cob = blockIdx.z;
  const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pp = i16 & 3;
  const bool has2 = wave + 8 < p.kw;                      // second dx slot of this wave is a real tap
  const bool dbwave = wave == 7;                          // its second tap slot (dx = 15) never holds a tap

  f32x4 acc[2][KH];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < KH; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // per-lane byte offset of a transposing read inside a line pair [c8 0][c8 1]: row q of the 4-pixel block, channels 4pp..
  const int lane_x = (((pp >> 1) * WG_XP + 8 * g + q) * 16) + (pp & 1) * 8;
  const int lane_d = (((pp >> 1) * WG_DP + 8 * g + q) * 16) + (pp & 1) * 8;
  const char* xbase = reinterpret_cast<const char*>(lds_x) + lane_x;
  const char* dbase = reinterpret_cast<const char*>(lds_d) + lane_d;
  const uint4* zsrc = bfx_zero + (lane & 15);
  const long xplane = (long)p.H * p.W, dplane = (long)p.OH * p.OW;
  bf16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (__bf16)1.0f;

  for (int strip = slice; strip < p.strips; strip += p.S) {
    const int b = strip / p.tilesX, tx = strip - b * p.tilesX;
    const int x0 = tx * 32;
    const uint4* xsb = p.xs + ((long)b * p.XC8 + cib * 2) * 2 * xplane;
    const uint4* dsb = p.dys + ((long)b * p.DC8 + cob * 2) * 2 * dplane;
    // X rows [row0, row0 + nrows) -> ring slots from slot0 (nrows % 4 == 0, no wrap inside a block)
    auto stage_x = [&](int slot0, int row0, int nrows) {
      const int ninstr = nrows * WG_XROW / 64;
      for (int i = wave; i < ninstr; i += 8) {
        const int e = i * 64 + lane;
        const int line = e / WG_XP, px = e - line * WG_XP;
        const int row = line >> 2, hl = (line >> 1) & 1, c8 = line & 1;
        const int gy = row0 + row, gx = x0 - p.pw + px;
        const bool ok = px < 32 + p.kw - 1 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W &&
                        cib * 2 + c8 < p.XC8;
        const long off = ((long)c8 * 2 + hl) * xplane + (long)gy * p.W + gx;
        glds16(ok ? xsb + off : zsrc, lds_x + slot0 * WG_XROW + i * 64);
      }
    };
    auto stage_d = [&](int buf, int row0) {
      for (int i = wave; i < WG_DG / 64; i += 8) {
        const int e = i * 64 + lane;
        const int line = e / WG_DP, px = e - line * WG_DP;
        const int row = line >> 2, hl = (line >> 1) & 1, c8 = line & 1;
        const int gy = row0 + row, gx = x0 + px;
        const bool ok = px < 32 && gy < p.OH && gx < p.OW && cob * 2 + c8 < p.DC8;
        const long off = ((long)c8 * 2 + hl) * dplane + (long)gy * p.OW + gx;
        glds16(ok ? dsb + off : zsrc, lds_d + buf * WG_DG + i * 64);
      }
    };
    __builtin_amdgcn_s_barrier();                      // the previous strip's last reads are done
    stage_x(0, -p.ph - 2, 24);                         // slot(row) = (row + ph + 2) & 31: rows -ph-2 .. -ph+21 in slots 0 .. 23
    stage_d(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int t = 0; t < p.steps; ++t) {
      const int y0 = t * WG_R;
      if (t + 1 < p.steps) {                           // next step's rows stream in behind this step's MFMAs
        stage_x((8 * t + 24) & 31, y0 - p.ph + 22, 8);
        stage_d((t + 1) & 1, y0 + WG_R);
      }
      const char* db_ = dbase + (t & 1) * (WG_DG * 16);
      bf16x8 ah[WG_R], al[WG_R];
#pragma unroll
      for (int y = 0; y < WG_R; ++y) {
        ah[y] = bfx_tr_frag(db_ + (y * WG_DROW) * 16);
        al[y] = bfx_tr_frag(db_ + (y * WG_DROW + 2 * WG_DP) * 16);
      }
#pragma unroll
      for (int sd = 0; sd < 2; ++sd) {
        if (sd == 1 && !has2) {
          if (dbwave && cib == 0) {                    // bias gradient: dY (hi + lo) times ones, all columns equal
#pragma unroll
            for (int y = 0; y < WG_R; ++y) {
              acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[y], ones, acc[1][0], 0, 0, 0);
              acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[y], ones, acc[1][0], 0, 0, 0);
            }
          }
          continue;
        }
        if (sd == 0 && wave >= p.kw) continue;
        const int dx = wave + 8 * sd;
        const char* xb = xbase + dx * 16;
#pragma unroll
        for (int d = 0; d < WG_R + KH - 1; ++d) {
          const int slot = (8 * t + 2 + d) & 31;       // ring slot of input row y0 - ph + d (wave-uniform)
          const char* xr = xb + slot * (WG_XROW * 16);
          const bf16x8 bh = bfx_tr_frag(xr);
          const bf16x8 bl = bfx_tr_frag(xr + 2 * WG_XP * 16);
#pragma unroll
          for (int y = 0; y < WG_R; ++y) {
            const int dy = d - y;
            if (dy >= 0 && dy < KH) {
              acc[sd][dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[y], bh, acc[sd][dy], 0, 0, 0);
              acc[sd][dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[y], bl, acc[sd][dy], 0, 0, 0);
              acc[sd][dy] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[y], bh, acc[sd][dy], 0, 0, 0);
            }
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                    // next step's rows have landed; this step's reads are done
    }
  }

  // lane (n = i16, g) holds dW[co = 4g + r][ci = n] of each of its taps
  const int ci = cib * 16 + i16;
  float* wsb = p.ws + (long)slice * p.Cout * p.NtotP;
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
    const int dx = wave + 8 * sd;
    if (dx < p.kw && ci < p.Cin) {
#pragma unroll
      for (int dy = 0; dy < KH; ++dy)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = cob * 16 + 4 * g + r;
          if (co < p.Cout) wsb[(long)co * p.NtotP + ((long)ci * KH + dy) * p.kw + dx] = acc[sd][dy][r];
        }
    }
  }
  if (dbwave && cib == 0 && i16 == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = cob * 16 + 4 * g + r;
      if (co < p.Cout) wsb[(long)co * p.NtotP + p.NtotP - 1] = acc[1][0][r];
    }
  }
}

// ws [S][Cout][NtotP] -> dw [Cout][Ntot] (+ db [Cout] from the last column); eight independent chains, fixed order
__global__ void bfx_reduce_partials_kernel(const float* __restrict__ ws, float* __restrict__ dw, float* __restrict__ db,
                                           int Cout, int Ntot, int NtotP, int S) {
  const long n = (long)Cout * NtotP;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
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
    const float sum = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    const int co = (int)(i / NtotP), j = (int)(i - (long)co * NtotP);
    if (j < Ntot) dw[(long)co * Ntot + j] = sum;
    else if (db) db[co] = sum;
  }
}

struct BfxWgPlan { bool ok; int tilesX, strips, S, steps, coBlocks, ciBlocks, NtotP; };

BfxWgPlan bfx_wg_plan(const mpa_conv_desc* d) {
  BfxWgPlan pl{};
  pl.ok = false;
  if (!d || (d->kh != 15 && d->kh != 9) || d->kw < 1 || d->kw > 15 || d->sh != 1 || d->sw != 1) return pl;
  const int OH = d->H + 2 * d->ph - d->kh + 1, OW = d->W + 2 * d->pw - d->kw + 1;
  if (OH <= 0 || OW <= 0 || d->ph < 0 || d->ph > d->kh - 1) return pl;
  pl.tilesX = (int)mpa_cdiv(OW, 32);
  pl.strips = d->B * pl.tilesX;
  pl.steps = (int)mpa_cdiv(OH, WG_R);
  pl.coBlocks = (int)mpa_cdiv(d->Cout, 16);
  pl.ciBlocks = (int)mpa_cdiv(d->Cin, 16);
  const int groups = pl.coBlocks * pl.ciBlocks;
  const int forceS = mpa_diag().bfx_wg_s;      // diagnostics
  int S = std::max(1, std::min(pl.strips, 256 / std::max(1, groups)));
  S = (int)mpa_cdiv(pl.strips, mpa_cdiv(pl.strips, S));                   // equal strips per slice
  if (forceS >= 1 && forceS <= pl.strips) S = forceS;
  pl.S = S;
  pl.NtotP = d->Cin * d->kh * d->kw + 1;
  pl.ok = true;
  return pl;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

int64_t mpa_bf16x3_split_bytes(int B, int C, int H, int W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return MPA_ERR_ARG;
  return (int64_t)B * mpa_cdiv(C, 8) * 2 * H * W * 16;
}

int mpa_bf16x3_split(const float* x, void* out, int B, int C, int H, int W, void* stream) {
  if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return MPA_ERR_ARG;
  const int C8 = (int)mpa_cdiv(C, 8);
  const long HW = (long)H * W, total = (long)B * C8 * HW;
  const int blocks = (int)std::min<long>(mpa_cdiv(total, 256), 256 * 16);
  MPA_LAUNCH(split_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (uint4*)out, C, C8, HW, total);
  return mpa_launch_status();
}

int mpa_conv2d_bf16x3_supported(const mpa_conv_desc* d, int mode) {
  if (mode == 2) return bfx_wg_plan(d).ok ? 1 : 0;
  const BfxGeom g = bfx_geom(d, mode);
  if (!g.ok) return 0;
  return bfx_plan(g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, 1, 1, g.ph, g.pw).ok ? 1 : 0;
}

int64_t mpa_conv2d_bf16x3_packed_bytes(const mpa_conv_desc* d, int mode) {
  const BfxGeom g = bfx_geom(d, mode);
  if (!g.ok) return MPA_ERR_UNSUPPORTED;
  const BfxPlan pl = bfx_plan(g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, 1, 1, g.ph, g.pw);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)pl.coTiles * pl.C8 * pl.QN * g.kh * 2 * 64 * 16;
}

int mpa_conv2d_bf16x3_pack(const mpa_conv_desc* d, int mode, const float* w, void* w_packed, void* stream) {
  if (!w || !w_packed) return MPA_ERR_ARG;
  const BfxGeom g = bfx_geom(d, mode);
  if (!g.ok) return MPA_ERR_UNSUPPORTED;
  const BfxPlan pl = bfx_plan(g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, 1, 1, g.ph, g.pw);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  BfxPackParams p{};
  p.w = w; p.wp = (uint4*)w_packed;
  p.Cout_w = d->Cout; p.Cin_w = d->Cin; p.kh = d->kh; p.kw = d->kw; p.mode = mode;
  p.CoutP = g.Cout; p.CinP = g.Cin; p.coTiles = pl.coTiles; p.nChunks = pl.C8; p.QN = pl.QN;
  p.total = (long)pl.coTiles * pl.C8 * pl.QN * g.kh * 2 * 64;
  const int blocks = (int)std::min<long>(mpa_cdiv(p.total, 256), 4096);
  MPA_LAUNCH(bfx_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  return mpa_launch_status();
}

int64_t mpa_conv2d_bf16x3_stats_rows(const mpa_conv_desc* d) {
  const BfxGeom g = bfx_geom(d, 0);
  if (!g.ok) return MPA_ERR_UNSUPPORTED;
  const BfxPlan pl = bfx_plan(g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, 1, 1, g.ph, g.pw);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)d->B * pl.tilesY * pl.tilesX;
}

int mpa_conv2d_bf16x3_fwd(const mpa_conv_desc* d, const void* xs, const void* w_packed, const float* bias, float* y, int act,
                          float slope, float* partials, void* stream) {
  if (!d || !xs || !w_packed || !y || d->B <= 0) return MPA_ERR_ARG;
  if (partials && act != MPA_ACT_NONE) return MPA_ERR_UNSUPPORTED;
  const BfxGeom g = bfx_geom(d, 0);
  if (!g.ok) return MPA_ERR_UNSUPPORTED;
  return bfx_launch(g, d->B, xs, w_packed, bias, y, act, slope, partials, (hipStream_t)stream);
}

int mpa_conv2d_bf16x3_bwd_data(const mpa_conv_desc* d, const void* dys, const void* w_packed, float* dx, void* stream) {
  if (!d || !dys || !w_packed || !dx || d->B <= 0) return MPA_ERR_ARG;
  const BfxGeom g = bfx_geom(d, 1);
  if (!g.ok) return MPA_ERR_UNSUPPORTED;
  if (g.H + 2 * g.ph - g.kh + 1 != d->H || g.W + 2 * g.pw - g.kw + 1 != d->W) return MPA_ERR_UNSUPPORTED;
  return bfx_launch(g, d->B, dys, w_packed, nullptr, dx, MPA_ACT_NONE, 0.f, nullptr, (hipStream_t)stream);
}

int64_t mpa_conv2d_bf16x3_bwd_weight_workspace(const mpa_conv_desc* d) {
  const BfxWgPlan pl = bfx_wg_plan(d);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)pl.S * d->Cout * pl.NtotP * 4;
}

int mpa_conv2d_bf16x3_bwd_weight(const mpa_conv_desc* d, const void* xs, const void* dys, float* dw, float* db,
                                 void* workspace, int64_t workspace_bytes, void* stream) {
  if (!d || !xs || !dys || !dw || d->B <= 0) return MPA_ERR_ARG;
  const BfxWgPlan pl = bfx_wg_plan(d);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < (int64_t)pl.S * d->Cout * pl.NtotP * 4) return MPA_ERR_WORKSPACE;
  BfxWgParams p{};
  p.xs = (const uint4*)xs; p.dys = (const uint4*)dys; p.ws = (float*)workspace;
  p.B = d->B; p.Cin = d->Cin; p.XC8 = (int)mpa_cdiv(d->Cin, 8); p.H = d->H; p.W = d->W;
  p.Cout = d->Cout; p.DC8 = (int)mpa_cdiv(d->Cout, 8);
  p.OH = d->H + 2 * d->ph - d->kh + 1; p.OW = d->W + 2 * d->pw - d->kw + 1;
  p.kw = d->kw; p.ph = d->ph; p.pw = d->pw;
  p.tilesX = pl.tilesX; p.strips = pl.strips; p.S = pl.S; p.steps = pl.steps; p.NtotP = pl.NtotP; p.ciBlocks = pl.ciBlocks;
  hipStream_t s = (hipStream_t)stream;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)conv_bfx_wgrad_kernel<15>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv_bfx_wgrad_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  const size_t lds_bytes = (size_t)(WG_XG + 2 * WG_DG) * 16;
  const dim3 wgrid((unsigned)pl.S, (unsigned)pl.ciBlocks, (unsigned)pl.coBlocks);
  if (d->kh == 15) MPA_LAUNCH(conv_bfx_wgrad_kernel<15>, wgrid, dim3(512), lds_bytes, s, p);
  else MPA_LAUNCH(conv_bfx_wgrad_kernel<9>, wgrid, dim3(512), lds_bytes, s, p);
  int rc = mpa_launch_status();
  if (rc) return rc;
  const int Ntot = pl.NtotP - 1;
  const long n = (long)d->Cout * pl.NtotP;
  MPA_LAUNCH(bfx_reduce_partials_kernel, dim3((unsigned)std::min<long>(mpa_cdiv(n, 256), 2048)), dim3(256), 0, s,
             (const float*)workspace, dw, db, d->Cout, Ntot, pl.NtotP, pl.S);
  return mpa_launch_status();
}

}  // extern "C"
