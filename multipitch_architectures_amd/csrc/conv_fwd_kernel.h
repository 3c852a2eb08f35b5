// The forward / backward-data kernel of conv_fwd.hip with its launch ladder, as a header: the instantiations are spread over
// conv_fwd_nb12.hip (16- / 32-cout tiles), conv_fwd_nb45.hip (64 / 80) and conv_fwd_nb36.hip (the 48- / 96-cout phase-store
// builds), so that the three compile side by side (one translation unit took four minutes) and an edit of one group cannot
// re-allocate the registers of another.  conv_fwd.hip keeps the planner calls, the filter packers and the C ABI.
#pragma once
#include "mpa_common.h"
#ifndef MPA_COMMON_CDIV
#define MPA_COMMON_CDIV 1
#endif
#include "conv_plan.h"
#include "conv_stage.h"
#include "conv_internal.h"

#include "conv_fwd_params.h"

namespace {

// ------------------------------------------------------------------------------------------------ forward kernel
// PH (phase stores): backward-data variants whose couts are (channel, x/y phase) pairs -- a separate instantiation, as
// EF is: the 16->128 forward sits on a register cliff and lost 10 % whenever either was compiled into the common kernel
// MINW = 2: built to at most 256 registers (two waves per SIMD) -- only for <2,12,9>, whose natural allocation (256 VGPRs + 23
// AGPRs) leaves one workgroup per CU: with 26 spilled registers and two resident workgroups the 64 -> 32 9x9 forward at batch 256
// runs at 115 instead of 104 TFLOP/s; small grids (local batch 32) keep the unspilled build, which is faster there.
template <int NB, int PB, int KW = 0, bool EF = false, bool PH = false, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void conv_fwd_kernel(const ConvFwdParams p) {
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
    const bool do_stage = (MPA_DBG(p) != 1 && MPA_DBG(p) != 3) || c == c_begin;
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
        if (MPA_DBG(p) != 2)
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
      if (MPA_DBG(p) != 2)
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
      if (dy + 1 < p.kh && MPA_DBG(p) != 3) {
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
        // (phase stores: the bias belongs to the channel the row maps to -- cout remainder fold of the forward pass)
        if (p.bias && blockIdx.z == 0) v += p.bias[PH ? co / (p.outXmul * p.outYmul) : co];
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

template <int NB, int PB, int KW, bool EF, bool PH = false, int MINW = 1>
int launch_fwd_ef(const FwdPlan& pl, const ConvFwdParams& p, dim3 grid, hipStream_t s) {
  static bool big_lds = false;
  if (!big_lds) {
    (void)hipFuncSetAttribute((const void*)conv_fwd_kernel<NB, PB, KW, EF, PH, MINW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              80 * 1024);
    big_lds = true;
  }
  MPA_LAUNCH((conv_fwd_kernel<NB, PB, KW, EF, PH, MINW>), grid, dim3(256), pl.lds_bytes, s, p);
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
  if constexpr (NB == 2 && PB == 12 && KW == 9) {      // two resident workgroups once the grid has work for them (MINW above)
    if ((long)grid.x * grid.z >= 1024 && 2 * pl.lds_bytes <= 160 * 1024 && !mpa_diag().fwd_no_minw)
      return launch_fwd_ef<NB, PB, KW, false, false, 2>(pl, p, grid, s);
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

// the plan fields the launch ladder reads, rebuilt from what crosses the translation units
inline FwdPlan fwd_plan_of(const MpaFwdLaunch& L) {
  FwdPlan pl{};
  pl.NB = L.NB; pl.PB = L.PB; pl.KWS = L.KWS; pl.coTiles = L.coTiles; pl.nChunks = L.nChunks; pl.lds_bytes = L.lds_bytes;
  pl.ok = true;
  return pl;
}

}  // namespace
