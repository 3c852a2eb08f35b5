// ------------------------------------------------------------------------------------------------ backward-weight, 15x15
// 88 % of the model's conv FLOPs sit in 15x15 stride-1 filters (inc, down1, upconv4; DRCNN prefilters), so their
// weight gradient gets a dedicated kernel: the MFMA N dimension is the 15 dx taps (padded to 16) of one (ci, dy) row
// and the X tile has a fixed LDS row pitch of 128 words, so every B-operand read is `base + immediate` (dy*512 B)
// -- one address VGPR for 15 reads instead of a running pointer per tap block.
//   wave tile: NBC cout blocks x CIW input channels x 15 dy  (acc = NBC*CIW*15 tiles of 16x16)
//   block    : 4 waves = 4*CIW input channels sharing one dY tile of NBC*16 couts
#include "mpa_common.h"
#define MPA_COMMON_CDIV 1
#include "conv_plan.h"
#include "conv_stage.h"
#include "conv_internal.h"

namespace {

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
    if (MPA_DBG(p) != 3) __syncthreads();
    if ((MPA_DBG(p) != 1 && MPA_DBG(p) != 3) || tile == split) {
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
    if (MPA_DBG(p) != 3 || tile == split) __syncthreads();
    if (MPA_DBG(p) == 2) continue;
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
    if (MPA_DBG(p) != 1 || tile == split)
      glds_stage_x16(lds_x, p.x + (long)b * p.Cin * p.H * p.W, lane, wave, 4, p.IH, W15_PITCH, xchp, p.TX64, ci_first,
                     oy0 - 7, ox0 - 8, p.Cin, p.H, p.W);
    float4 an[W15G_DEPTH][NBC];       // dY quads of the next W15G_DEPTH groups, in flight
#pragma unroll
    for (int d = 0; d < W15G_DEPTH; ++d) load_full(an[d], d / nfull, d % nfull);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (MPA_DBG(p) == 2) continue;

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

}  // namespace

extern "C" {

int64_t mpa_conv2d_bwd_weight_workspace(const mpa_conv_desc* d) {
  if (!d) return MPA_ERR_ARG;
  if (plan_head_wgrad(d).ok) return mpa_conv_head_wgrad_workspace(d);
  Wg15Plan p15 = plan_wgrad15(d);
  if (p15.ok) return (int64_t)p15.S * d->Cout * (d->Cin * 225 + 1) * 4;
  WgPlan pl = plan_wgrad(d);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)pl.S * d->Cout * (pl.Ntot + 1) * 4;
}

int mpa_conv2d_bwd_weight(const mpa_conv_desc* d, const float* x, const float* dy, float* dw, float* db,
                          void* workspace, int64_t workspace_bytes, void* stream) {
  if (!d || !x || !dy || !dw || d->B <= 0) return MPA_ERR_ARG;
  if (plan_head_wgrad(d).ok)
    return mpa_conv_head_bwd_weight(d, x, dy, dw, db, workspace, workspace_bytes, (hipStream_t)stream);
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
    q.dbg = mpa_diag().dbg_wg15;
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
    return mpa_conv_reduce_partials((const float*)workspace, dw, db, d->Cout, Ntot, Ntot + 1, p15.S, s15);
  }
  return mpa_conv_wgrad_generic(d, x, dy, dw, db, workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
