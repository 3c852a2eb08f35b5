// Head conv2 of every model: nn.Conv2d(n0, n1, (3,3), stride (1,3), padding (1,0))  (unet_cnns.py:538-543,
// basic_cnns.py:390-395), exact fp32 on v_mfma_f32_16x16x4_f32.
//
// The column stride equals the filter width: output pixel p = oy*OW + ox reads the flattened input plane at
// 3p + (dy-1)*W + dx, which is linear in p (conv_plan.h: plan_head).  So all three passes are GEMMs whose B operand is a
// shifted view of one contiguous slab per channel:
//   * forward / backward-data share head_gemm_kernel: a workgroup owns PXT consecutive pixels of one image and every
//     output row (couts; or (channel, column phase) pairs for backward-data); a wave owns MT 16-row tiles x 5 pixel blocks,
//     so one A read + one B read feed 5*MT/(5+MT) MFMAs each.  Channel chunks of the slab and of the packed filters stream
//     in by LDS-DMA behind the MFMAs of the previous chunk (double-buffered, one barrier per chunk).
//   * head_wgrad_kernel: a workgroup (4 waves x 16 input channels, all 9 taps x MT cout tiles in registers) walks a range
//     of 64-pixel chunks; per chunk the dY slab once and one X slab per filter row -- the row of tap row dy is the
//     contiguous run 3p + (dy-1)W.., so there is no halo to stage -- and writes one partial result per slice; slices are
//     reduced in a fixed order (conv_wgrad.hip: mpa_conv_reduce_partials).
// Image rows above / below the plane come from a zero page.  LDS bank conflicts: the slab pitch is == 16 (mod 32) words,
// which makes the forward's stride-3 B reads conflict-free; the backward-weight reads are 2-way conflicted at 8 reads per
// 15 MFMAs, i.e. irrelevant.
#define MPA_COMMON_CDIV 1
#include "mpa_common.h"
#include "conv_plan.h"
#include "conv_internal.h"
#include <type_traits>

namespace {

__device__ __attribute__((aligned(16))) uint4 head_zero[64];



__device__ __forceinline__ void hglds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct HeadParams {
  const float* src;      // forward: x [B][K][SL]; backward-data: dy [B][K][SL]
  const float* wp;       // packed A operand [chunk][AUw]
  const float* bias;
  float* out;
  int K, Mrows, MTT, P, SL, HALO, XS, CK, nChunks, tilesP, xinstr, xmagic;
  int NG, grp_shift;     // tap groups per channel chunk (tall filters: chunk = (channel chunk, tap group)); slab origin shift per group
  int used;              // words of a slab row that come from the plane
  int xslab;             // words of one X slab buffer (multiple of 256)
  int AUw;
  int tapoff[25];
  long srcBS, outBS;     // batch strides (words)
  long outPS;            // backward-data: words of one dx plane (H*W)
  int act;
  float slope;
  int dbg;               // MPA_HEAD_DEBUG (timing experiments, wrong results): 2 stage only the first chunk, 3 no MFMAs
};


// MODE 0: conv2 forward (source stride 3 per pixel, rows = couts); 1: conv2 backward-data (rows = (channel, column phase), phase
// stores); 2: tall (kh, 1) filters -- conv3 at T > 75, forward and backward-data (conv_plan.h: plan_tall): rows = output
// channels, the kh taps in groups of NT, a chunk = (channel chunk, tap group) whose slab is the group's window of the plane
template <int MT, int WM, int WN, int NT, int MODE>
__global__ __launch_bounds__(WM * WN * 64) void head_gemm_kernel(const HeadParams p) {
  extern __shared__ __attribute__((aligned(16))) float hlds[];
  constexpr bool FWD = MODE != 1;
  constexpr int NB = 5, NW = WM * WN, PXT = WN * NB * 16, SN = MODE == 0 ? 3 : 1;
  constexpr int NSO = 12;      // X staging pieces per wave whose pattern is tabulated
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int j = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x % p.tilesP, b = blockIdx.x / p.tilesP;
  const int p0 = tile * PXT;
  // buffer b of the X slab at hlds + b * xslab, of the filter chunk at hlds + 2 * xslab + b * AUw (offsets, not a pointer
  // table: a table of LDS pointers decays to generic pointers and every operand read becomes a flat load)

  const float* srcb = p.src + (long)b * p.srcBS;
  const uint4* zsrc = head_zero + lane;
  // Staging: one 64-lane DMA instruction ("piece") per 1 KiB.  Unit u (16 bytes) of a chunk's slab -> channel c = u / (XS/4)
  // (multiply-shift, checked on the host) and word o of that channel's slab row; what piece n of this wave moves is the same
  // for every chunk (tabulated in so[]), the window of the plane it comes from depends on the chunk's tap group.
  // (measured: issuing the pieces one per tap step inside the MFMA stream was slower than the burst after the barrier --
  // the extra live addresses cost registers and the per-step branches issue slots: 106.7 against 111.8 TFLOP/s)
  constexpr bool INTER = false;
  // so[n] = (channel of the chunk << 16) | word of the slab row, -1 = pitch padding / nothing to move
  int so[NSO];
  const int xu4 = p.XS >> 2;
#pragma unroll
  for (int n = 0; n < NSO; ++n) {
    const int i = wave + n * NW, u = i * 64 + lane;
    const int c = (int)(((unsigned)u * (unsigned)p.xmagic) >> 20), o = (u - c * xu4) * 4;
    so[n] = (i < p.xinstr && c < p.CK && o < p.used) ? (c << 16) | o : -1;
  }
  const int nxw = (p.xinstr + NW - 1 - wave) / NW, naw = ((p.AUw >> 8) + NW - 1 - wave) / NW;   // pieces of this wave
  // (qg: plane word of the slab's first word for this chunk's tap group; sc: the chunk's first channel; kleft: channels left)
  auto xpiece = [&](int n, int off, int chunk, int buf, int qg, const float* sc, int kleft) {
    if (n < nxw && !(MPA_DBG(p) == 4 && chunk)) {
      const int c = off >> 16, q = qg + (off & 0xffff);
      // channels past K (last chunk) and words outside the plane (rows above / below it) are zeros
      const bool ok = off >= 0 && c < kleft && q >= 0 && q < p.SL;
      hglds16(ok ? (const void*)(sc + (long)c * p.SL + q) : (const void*)zsrc, hlds + buf * p.xslab + (wave + n * NW) * 256);
    }
  };
  auto apiece = [&](int n, int chunk, int buf) {
    if (n < naw && !(MPA_DBG(p) == 5 && chunk))
      hglds16(reinterpret_cast<const uint4*>(p.wp) + (long)chunk * (p.AUw >> 2) + (wave + n * NW) * 64 + lane,
              hlds + 2 * p.xslab + buf * p.AUw + (wave + n * NW) * 256);
  };
  // everything of a chunk at once
  // MODE 0 / 1: what a piece moves does not depend on the chunk beyond a uniform channel offset, so the pieces go out as buffer
  // loads to LDS -- per-lane byte offset from a table (a large offset for zeros: the resource's bounds check returns 0 for rows
  // above / below the plane, pitch padding and channels past K), the chunk's offset on the scalar unit: ~3 instructions per
  // piece instead of ~10 (measured: the staging burst costs the single wave of a SIMD its MFMA slots)
  constexpr bool FAST = MODE != 2;
  int vo[NSO];
  if constexpr (FAST) {
#pragma unroll
    for (int n = 0; n < NSO; ++n) {
      const int c = so[n] >> 16, q = SN * p0 - p.HALO + (so[n] & 0xffff);
      vo[n] = (so[n] >= 0 && q >= 0 && q < p.SL) ? (c * p.SL + q) * 4 : 0x40000000;
    }
  }
  const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)srcb, 0, (int)min((long)p.K * p.SL * 4, 0x3fffffffL), 0x00020000);
  const auto arsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, (int)min((long)p.nChunks * p.AUw * 4, 0x3fffffffL), 0x00020000);
  auto stage_fast = [&](int chunk, int buf) {
    const int xso = chunk * p.CK * p.SL * 4, aso = chunk * p.AUw * 4;
    if (!(MPA_DBG(p) == 4 && chunk)) {
#pragma unroll
      for (int n = 0; n < NSO; ++n)
        if (n < nxw)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(hlds + buf * p.xslab + (wave + n * NW) * 256),
                                                   16, vo[n], xso, 0, 0);
    }
    if (!(MPA_DBG(p) == 5 && chunk)) {
#pragma unroll 1
      for (int n = 0; n < naw; ++n)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (__attribute__((address_space(3))) void*)(hlds + 2 * p.xslab + buf * p.AUw + (wave + n * NW) * 256),
                                                 16, ((wave + n * NW) * 64 + lane) * 16, aso, 0, 0);
    }
  };
  auto stage_from = [&](int n0, int chunk, int buf) {
    if constexpr (FAST) {           // (head_params refuses tiles with more than NSO pieces per wave)
      stage_fast(chunk, buf);
      return;
    }
    int cq = chunk, tg = 0;
    if (MODE == 2) { cq = chunk / p.NG; tg = chunk - cq * p.NG; }
    const int qg = SN * p0 - p.HALO + tg * p.grp_shift, kleft = p.K - cq * p.CK;
    const float* sc = srcb + (long)cq * p.CK * p.SL;
#pragma unroll
    for (int n = 0; n < NSO; ++n)
      if (n >= n0) xpiece(n, so[n], chunk, buf, qg, sc, kleft);
#pragma unroll 1
    for (int n = max(n0, NSO); n < nxw; ++n) {          // (more X pieces than the table holds: not with the planners' tiles)
      const int u = (wave + n * NW) * 64 + lane;
      const int c = (int)(((unsigned)u * (unsigned)p.xmagic) >> 20), o = (u - c * xu4) * 4;
      xpiece(n, (c < p.CK && o < p.used) ? (c << 16) | o : -1, chunk, buf, qg, sc, kleft);
    }
#pragma unroll 1
    for (int n = n0; n < naw; ++n) apiece(n, chunk, buf);
  };

  f32x4 acc[MT][NB];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[t][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // tall backward-data: dx pixel q takes dY[q - dy W], which exists for dy in [q / W - (OH - 1), q / W] only -- tap groups
  // that fall outside that interval for every pixel of the tile are skipped (57 % of the taps exist for T = 174)
  int tg_lo = 0, ng = p.NG;
  if (MODE == 2 && p.grp_shift < 0) {
    const int W = -p.grp_shift / NT, OHm1 = p.SL / W - 1;
    const int dy_hi = min(p.NG * NT - 1, (p0 + PXT - 1) / W), dy_lo = max(0, p0 / W - OHm1);
    tg_lo = dy_lo / NT;
    ng = max(1, dy_hi / NT - tg_lo + 1);
  }
  const int nch = (p.nChunks / p.NG) * ng;
  auto real = [&](int i) { const int cq = i / ng; return cq * p.NG + tg_lo + (i - cq * ng); };   // i-th chunk this tile runs
  stage_from(0, real(0), 0);
  for (int ci = 0; ci < nch; ++ci) {
    const int buf = ci & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();              // this chunk has landed everywhere; nobody still reads the other buffer
    const bool more = ci + 1 < nch && MPA_DBG(p) < 2;
    if (MPA_DBG(p) == 3 || !INTER) {
      if (more) stage_from(0, real(ci + 1), buf ^ 1);
      if (MPA_DBG(p) == 3) continue;
    }
    const float* xb = hlds + buf * p.xslab + g * p.XS + SN * (wn * NB * 16 + j);
    const float* ab = hlds + 2 * p.xslab + buf * p.AUw + wm * MT * 64 + lane;
    // the operands of step (kq, tap) + 1 are requested before the MFMAs of step (kq, tap) issue: with one wave per SIMD
    // nothing else hides the LDS latency (left to the compiler, the reads sank to the end of each MFMA group: 70 % MFMA rate)
    const int nkq = p.CK >> 2;
    float a0[MT], b0[NB];
#pragma unroll
    for (int t = 0; t < MT; ++t) a0[t] = ab[t * 64];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) b0[nb] = xb[SN * 16 * nb + p.tapoff[0]];
    for (int kq = 0; kq < nkq; ++kq) {
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) {
        float a1[MT], b1[NB];
        // (the compiler's s_waitcnt in front of a step's first MFMA is lgkmcnt(0): it also waits for reads issued just before
        // it, so the next step's reads go out after this step's first row of MFMAs and have the other rows to land)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], b0[nb], acc[0][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        {   // step + 1, clamped to the chunk's last step (whose operands are then simply read twice)
          const int kn = tap + 1 < NT ? kq : min(kq + 1, nkq - 1), tn = tap + 1 < NT ? tap + 1 : (kq + 1 < nkq ? 0 : NT - 1);
#pragma unroll
          for (int t = 0; t < MT; ++t) a1[t] = ab[((kn * NT + tn) * p.MTT + t) * 64];
          const int toff = tap + 1 < NT ? p.tapoff[tap + 1 < NT ? tap + 1 : 0] : (kq + 1 < nkq ? p.tapoff[0] : p.tapoff[NT - 1]);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) b1[nb] = xb[kn * 4 * p.XS + SN * 16 * nb + toff];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 1; t < MT; ++t)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb) acc[t][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], b0[nb], acc[t][nb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < MT; ++t) a0[t] = a1[t];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) b0[nb] = b1[nb];
      }
    }
  }

  // lane (j, g) holds rows 4g .. 4g+3 of each 16-row tile at pixel column j of each pixel block
  float* ob = p.out + (long)b * p.outBS;
  if constexpr (MODE == 1 && MT % 3 == 0) {
    // backward-data: three 16-row tiles are 16 channels x 3 column phases, i.e. for a 16-pixel block 16 runs of 48 consecutive
    // words of dx.  Transposed through a wave-private LDS patch they go out as 16-byte stores (30 per lane instead of 120
    // scattered 4-byte ones: the scalar epilogue was ~10 % of the workgroup's life).
    __syncthreads();                                  // the main loop's LDS images are dead now
    float* patch = hlds + wave * (48 * 17);           // [row 48][pixel 16 (+1 pad)]
#pragma unroll
    for (int grp = 0; grp < MT / 3; ++grp) {
      const int ci0 = ((wm * MT + 3 * grp) * 16) / 3;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int tl = 0; tl < 3; ++tl)
#pragma unroll
          for (int r = 0; r < 4; ++r) patch[(tl * 16 + 4 * g + r) * 17 + j] = acc[3 * grp + tl][nb][r];
        const int pxb = p0 + (wn * NB + nb) * 16;
        const int nvalid = 3 * min(16, p.P - pxb);    // words of each run that exist (a multiple of 4)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int idx = lane + 64 * k, cl = idx / 12, f4 = idx - cl * 12;
          float4 v;
          float* e = reinterpret_cast<float*>(&v);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int wd = 4 * f4 + q, px = wd / 3, ph = wd - 3 * px;
            e[q] = patch[(cl * 3 + ph) * 17 + px];
          }
          if (ci0 + cl < p.Mrows / 3 && 4 * f4 < nvalid)
            *reinterpret_cast<float4*>(ob + (long)(ci0 + cl) * p.outPS + 3 * (long)pxb + 4 * f4) = v;
        }
      }
    }
    return;
  }
  if constexpr (FWD) {
    // forward (and the tall filters): every 16 x 16 accumulator tile through a wave-private LDS patch so that a lane owns 4
    // consecutive pixels of one output row and writes one 16-byte store (25-35 per lane instead of 100-140 4-byte ones)
    // Bias and activation are applied in the accumulator layout, from bias values fetched before the barrier: a load that is
    // first used inside the divergent store blocks is waited for in every one of them (vmcnt(0): one in-order counter for
    // loads and stores), which made each store wait for the one before it.
    float bs[MT][4];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) bs[t][r] = p.bias ? p.bias[min((wm * MT + t) * 16 + 4 * g + r, p.Mrows - 1)] : 0.f;
    __syncthreads();                                  // the main loop's LDS images are dead now
    float* patch = hlds + wave * (16 * 20);           // [row 16][pixel 16 (+4 pad)]
    const int rl = lane >> 2, quad = lane & 3;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = (wm * MT + t) * 16 + rl;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) patch[(4 * g + r) * 20 + j] = mpa_apply_act(acc[t][nb][r] + bs[t][r], p.act, p.slope);
        const float4 v = *reinterpret_cast<const float4*>(patch + rl * 20 + 4 * quad);
        const int px = p0 + (wn * NB + nb) * 16 + 4 * quad;
        if (m < p.Mrows && px < p.P)                  // (P % 4 == 0: a quad is inside or outside as a whole)
          *reinterpret_cast<float4*>(ob + (long)m * p.P + px) = v;
      }
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = (wm * MT + t) * 16 + 4 * g + r;
      if (m >= p.Mrows) continue;
      if (FWD) {
        const float bs = p.bias ? p.bias[m] : 0.f;
        float* orow = ob + (long)m * p.P;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int px = p0 + (wn * NB + nb) * 16 + j;
          if (px < p.P) orow[px] = mpa_apply_act(acc[t][nb][r] + bs, p.act, p.slope);
        }
      } else {
        const int ci = m / 3, ph = m - 3 * ci;
        float* orow = ob + (long)ci * p.outPS + ph;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          const int px = p0 + (wn * NB + nb) * 16 + j;
          if (px < p.P) orow[3 * px] = acc[t][nb][r];
        }
      }
    }
}

template <int MT, int WM, int WN, int NT, int MODE>
int head_launch_one(const HeadPlan& pl, const HeadParams& p, int B, hipStream_t s) {
  auto k = head_gemm_kernel<MT, WM, WN, NT, MODE>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024); attr = true; }
  MPA_LAUNCH(k, dim3((unsigned)((long)B * pl.tilesP)), dim3(WM * WN * 64), pl.lds_bytes, s, p);
  return mpa_launch_status();
}

int head_launch(const HeadPlan& pl, const HeadParams& p, int B, hipStream_t s) {
#define MPA_HEAD_CASE(MT_, WM_, WN_, NT_, MODE_) \
  if (pl.MT == MT_ && pl.WM == WM_ && pl.WN == WN_ && pl.NT == NT_) return head_launch_one<MT_, WM_, WN_, NT_, MODE_>(pl, p, B, s)
  if (pl.mode == 0) {
    MPA_HEAD_CASE(4, 1, 4, 9, 0); MPA_HEAD_CASE(5, 1, 4, 9, 0); MPA_HEAD_CASE(6, 1, 4, 9, 0); MPA_HEAD_CASE(7, 1, 4, 9, 0);
    MPA_HEAD_CASE(4, 2, 4, 9, 0); MPA_HEAD_CASE(5, 2, 4, 9, 0); MPA_HEAD_CASE(6, 2, 4, 9, 0); MPA_HEAD_CASE(7, 2, 4, 9, 0);
  } else if (pl.mode == 1) {
    MPA_HEAD_CASE(4, 2, 4, 3, 1); MPA_HEAD_CASE(5, 2, 4, 3, 1); MPA_HEAD_CASE(6, 2, 4, 3, 1); MPA_HEAD_CASE(7, 2, 4, 3, 1);
    MPA_HEAD_CASE(5, 4, 2, 3, 1); MPA_HEAD_CASE(6, 4, 2, 3, 1); MPA_HEAD_CASE(7, 4, 2, 3, 1);
  } else {                      // tall filters, forward and backward-data
    MPA_HEAD_CASE(4, 1, 4, 25, 2); MPA_HEAD_CASE(5, 1, 4, 25, 2);
    MPA_HEAD_CASE(6, 1, 4, 15, 2); MPA_HEAD_CASE(7, 1, 4, 15, 2);
    MPA_HEAD_CASE(4, 2, 4, 15, 2); MPA_HEAD_CASE(5, 2, 4, 15, 2); MPA_HEAD_CASE(6, 2, 4, 15, 2); MPA_HEAD_CASE(7, 2, 4, 15, 2);
  }
#undef MPA_HEAD_CASE
  return MPA_ERR_UNSUPPORTED;
}

int head_params(const mpa_conv_desc* d, const HeadPlan& pl, HeadParams& p) {
  p.K = pl.K; p.Mrows = pl.Mrows; p.MTT = pl.MTT; p.P = pl.P; p.SL = pl.SL; p.HALO = pl.HALO; p.XS = pl.XS; p.CK = pl.CK;
  p.nChunks = pl.nChunks; p.tilesP = pl.tilesP;
  const int NW = pl.WM * pl.WN;
  const long instr = mpa_cdiv((long)pl.CK * (pl.XS / 4), 64);
  p.xinstr = (int)instr;
  if (pl.mode < 2 && mpa_cdiv(instr, NW) > 12) return MPA_ERR_UNSUPPORTED;      // head_gemm_kernel: NSO table entries per wave
  p.xslab = (int)(instr * 256);      // the last instruction ends inside the buffer
  const int xu4 = pl.XS / 4;
  p.xmagic = (int)(((1L << 20) + xu4 - 1) / xu4);
  for (long u = 0; u < instr * 64; ++u)
    if ((int)(((unsigned long)u * (unsigned long)p.xmagic) >> 20) != (int)(u / xu4) || u * (long)p.xmagic >= (1L << 32))
      return MPA_ERR_UNSUPPORTED;
  p.AUw = (int)pl.AUw;
  p.dbg = mpa_diag().dbg_head;
  for (int t = 0; t < 25; ++t) p.tapoff[t] = 0;
  p.NG = 1; p.grp_shift = 0;
  if (pl.mode == 0) {
    p.used = 3 * pl.PXT + 2 * pl.HALO;
    for (int dy = 0; dy < 3; ++dy) for (int dx = 0; dx < 3; ++dx) p.tapoff[dy * 3 + dx] = dy * d->W + dx;
  } else if (pl.mode == 1) {
    p.used = pl.PXT + 2 * pl.HALO;
    for (int dy = 0; dy < 3; ++dy) p.tapoff[dy] = (2 - dy) * (d->W / 3);
  } else {
    p.used = pl.PXT + (pl.NT - 1) * d->W;
    p.NG = d->kh / pl.NT;
    if (pl.mode == 2) { p.grp_shift = pl.NT * d->W; for (int t = 0; t < pl.NT; ++t) p.tapoff[t] = t * d->W; }
    else { p.grp_shift = -pl.NT * d->W; for (int t = 0; t < pl.NT; ++t) p.tapoff[t] = (pl.NT - 1 - t) * d->W; }
  }
  return MPA_OK;
}

// ------------------------------------------------------------------------------------------------ backward-weight
struct HeadWgParams {
  const float* x;
  const float* dy;
  float* ws;            // [S][Cout][Cin*9 + 1]
  int B, Cin, Cout, H, W, OW, P, HW, S;
  int NCS, SEG, KS, NRB, RB;
  int XPu, DPu, XUs, DUs, xmagic, dmagic;
  long items, itemsPer;
  int dbg;              // MPA_HEAD_DEBUG (timing experiments, wrong results): 2 stage nothing after an item's first rows, 3 no MFMAs
};

// KSC: k steps (4 pixels each) of a segment at compile time (9 for the 72-column rows of every model), 0 = p.KS at run time
constexpr int HEAD_WG_NX = 8, HEAD_WG_ND = 6;   // staging pieces per wave, at most: input-row slot, dY buffer (plan_head_wgrad)

template <int MT, int KSC>
__global__ __launch_bounds__(256) void head_wgrad_kernel(const HeadWgParams p) {
  extern __shared__ __attribute__((aligned(16))) uint4 wlds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int slice = blockIdx.x, cg = blockIdx.y, cog = blockIdx.z;
  const long it_first = (long)slice * p.itemsPer, it_last = min(p.items, it_first + p.itemsPer);
  uint4* const xs0 = wlds;                    // X ring slot r at xs0 + r * XUs, dY buffer b at ds0 + b * DUs
  uint4* const ds0 = wlds + 3 * p.XUs;
  const uint4* zsrc = head_zero + lane;
  const int XP = p.XPu * 4, DP = p.DPu * 4;   // pitches in words

  f32x4 acc[MT][9];
  float dbacc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    dbacc[t] = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[t][k] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // per-wave staging pattern (the same for every row of every item): word offset of piece n inside (image, row), -1 = zeros
  const int nx = ((p.XUs >> 6) + 3 - wave) >> 2, nd = ((p.DUs >> 6) + 3 - wave) >> 2;
  int xpo[HEAD_WG_NX], dpo[HEAD_WG_ND];
#pragma unroll
  for (int n = 0; n < HEAD_WG_NX; ++n) {
    const int u = (wave + 4 * n) * 64 + lane;
    const int c = (int)(((unsigned)u * (unsigned)p.xmagic) >> 20), o = u - c * p.XPu;
    const int ch = cg * 64 + c;
    xpo[n] = (n < nx && c < 64 && o < 3 * p.KS && ch < p.Cin) ? ch * p.HW + 4 * o : -1;      // 3 KS units per row segment
  }
#pragma unroll
  for (int n = 0; n < HEAD_WG_ND; ++n) {
    const int u = (wave + 4 * n) * 64 + lane;
    const int row = (int)(((unsigned)u * (unsigned)p.dmagic) >> 20), o = u - row * p.DPu;
    const int co = cog * MT * 16 + row;
    dpo[n] = (n < nd && row < 16 * MT && o < p.KS && co < p.Cout) ? co * p.P + 4 * o : -1;
  }

  for (long item = it_first; item < it_last; ++item) {
    const int rb = (int)(item % p.NRB), sg = (int)((item / p.NRB) % p.NCS), b = (int)(item / ((long)p.NRB * p.NCS));
    const int r0 = rb * p.RB, r1 = min(p.H, r0 + p.RB);
    const float* xb = p.x + (long)b * p.Cin * p.HW + 3 * sg * p.SEG;
    const float* db = p.dy + (long)b * p.Cout * p.P + sg * p.SEG;
    // staging, one 64-lane DMA instruction ("piece") at a time so that it can be issued between MFMAs: pieces 0 .. nx-1 of a
    // wave move input row y (segment) of the 64 channels into a ring slot (rows outside the image are zeros), pieces
    // nx .. nx+nd-1 move dY row h.  What a piece moves within a row is the same for every row: xpo / dpo (set up below).
    auto piece = [&](int n, int y, int slot, int h, int dbuf) {
      if (n < HEAD_WG_NX) {
        if (n < nx) {
          const bool ok = xpo[n < HEAD_WG_NX ? n : 0] >= 0 && y >= 0 && y < p.H;
          hglds16(ok ? (const void*)(xb + xpo[n < HEAD_WG_NX ? n : 0] + y * p.W) : (const void*)zsrc,
                  xs0 + slot * p.XUs + (wave + 4 * n) * 64);
        }
      } else if (n - HEAD_WG_NX < nd) {
        const int m = n - HEAD_WG_NX;
        const bool ok = dpo[m < HEAD_WG_ND ? m : 0] >= 0;
        hglds16(ok ? (const void*)(db + dpo[m < HEAD_WG_ND ? m : 0] + h * p.OW) : (const void*)zsrc,
                ds0 + dbuf * p.DUs + (wave + 4 * m) * 64);
      }
    };
    auto stage_x = [&](int y, int slot) {
#pragma unroll
      for (int n = 0; n < HEAD_WG_NX; ++n) piece(n, y, slot, 0, 0);
    };
    auto stage_d = [&](int h, int dbuf) {
#pragma unroll
      for (int n = 0; n < HEAD_WG_ND; ++n) piece(HEAD_WG_NX + n, 0, 0, h, dbuf);
    };
    const int KS = KSC ? KSC : p.KS;
    // one filter row: acc[.][3 dyr + dx] += dY(row h) x X(ring slot), K = the segment's pixels, 4 per MFMA.  pc0 >= 0: k step ks
    // also issues staging piece pc0 + ks of (input row sy -> slot ss, dY row sh -> buffer sd) in the shadow of its MFMAs.
    auto taps = [&](auto dyr_c, auto pc0_c, bool stg, int slot, int dbuf, int sy, int ss, int sh, int sd) {
      constexpr int dyr = decltype(dyr_c)::value, pc0 = decltype(pc0_c)::value;
      const float* xf = reinterpret_cast<const float*>(xs0 + slot * p.XUs) + (16 * wave + j) * XP + 3 * g;
      const float* df = reinterpret_cast<const float*>(ds0 + dbuf * p.DUs) + j * DP + g;
      float a0[MT], x0[3];
#pragma unroll
      for (int t = 0; t < MT; ++t) a0[t] = df[t * 16 * DP];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) x0[dx] = xf[dx];
      auto step = [&](int ks) {
        float a1[MT], x1[3];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          acc[0][dyr * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[0], x0[dx], acc[0][dyr * 3 + dx], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        {   // next k step's operands behind this step's first MFMAs (see head_gemm_kernel); the last step re-reads its own
          const int kn = min(ks + 1, KS - 1);
#pragma unroll
          for (int t = 0; t < MT; ++t) a1[t] = df[t * 16 * DP + kn * 4];
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) x1[dx] = xf[kn * 12 + dx];
        }
        if (dyr == 0) {
#pragma unroll
          for (int t = 0; t < MT; ++t) dbacc[t] += a0[t];
        }
        if constexpr (KSC > 0) {
          if (stg) piece(pc0 + ks, sy, ss, sh, sd);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 1; t < MT; ++t)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
            acc[t][dyr * 3 + dx] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t], x0[dx], acc[t][dyr * 3 + dx], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < MT; ++t) a0[t] = a1[t];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) x0[dx] = x1[dx];
      };
      if constexpr (KSC > 0) {
#pragma unroll
        for (int ks = 0; ks < KSC; ++ks) step(ks);
      } else {
#pragma unroll 1
        for (int ks = 0; ks < KS; ++ks) step(ks);
      }
    };

    __builtin_amdgcn_s_barrier();               // the previous item's last reads are done: the ring is free
    stage_x(r0 - 1, 0);
    stage_x(r0, 1);
    stage_x(r0 + 1, 2);
    stage_d(r0, 0);
    for (int h = r0, i = 0; h < r1; ++h, ++i) {
      const int s0 = i % 3, s1 = (i + 1) % 3, s2 = (i + 2) % 3, dbuf = i & 1;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();             // input row h + 1 and dY row h have landed
      if (MPA_DBG(p) != 3) taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, false, s0, dbuf, 0, 0, 0, 0);
      __builtin_amdgcn_s_barrier();             // everyone is done with input row h - 1: its slot takes row h + 2 ...
      const bool more = h + 1 < r1 && MPA_DBG(p) < 2;
      if (MPA_DBG(p) != 3) {                         // ... piece by piece behind the MFMAs of the other two filter rows
        taps(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, more, s1, dbuf, h + 2, s0, h + 1, dbuf ^ 1);
        taps(std::integral_constant<int, 2>{}, std::integral_constant<int, KSC>{}, more, s2, dbuf, h + 2, s0, h + 1, dbuf ^ 1);
      }
      if (more) {
#pragma unroll
        for (int n = 0; n < HEAD_WG_NX + HEAD_WG_ND; ++n)
          if (n >= 2 * KSC || MPA_DBG(p) == 3) piece(n, h + 2, s0, h + 1, dbuf ^ 1);
      }
    }
  }

  // D[co = 4g + r][ci = j]
  const int Ntot = p.Cin * 9, NtotP = Ntot + 1;
  float* wsl = p.ws + (long)slice * p.Cout * NtotP;
  const int ci = cg * 64 + wave * 16 + j;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = (cog * MT + t) * 16 + 4 * g + r;
      if (co < p.Cout && ci < p.Cin) {
        float* o = wsl + (long)co * NtotP + ci * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = acc[t][k][r];
      }
    }
    float v = dbacc[t];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int co = (cog * MT + t) * 16 + j;
    if (cg == 0 && wave == 0 && g == 0 && co < p.Cout) wsl[(long)co * NtotP + Ntot] = v;
  }
}

template <int MT>
int head_wgrad_launch(const HeadWgPlan& pl, const HeadWgParams& p, hipStream_t s) {
  auto k = (p.KS == 9 && !mpa_diag().head_wg_rolled) ? head_wgrad_kernel<MT, 9> : head_wgrad_kernel<MT, 0>;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)head_wgrad_kernel<MT, 9>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute((const void*)head_wgrad_kernel<MT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    attr = true;
  }
  MPA_LAUNCH(k, dim3((unsigned)pl.S, (unsigned)pl.chGroups, (unsigned)pl.coGroups), dim3(256), pl.lds_bytes, s, p);
  return mpa_launch_status();
}

}  // namespace

int mpa_conv_head_fwd(const mpa_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, float slope,
                      hipStream_t s) {
  const HeadPlan pl = plan_head(d, 0);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  HeadParams p{};
  const int rc = head_params(d, pl, p);
  if (rc) return rc;
  p.src = x; p.wp = wp; p.bias = bias; p.out = y;
  p.srcBS = (long)d->Cin * pl.SL; p.outBS = (long)d->Cout * pl.P; p.outPS = 0;
  p.act = act; p.slope = slope;
  return head_launch(pl, p, d->B, s);
}

int mpa_conv_head_bwd_data(const mpa_conv_desc* d, const float* dy, const float* wp, float* dx, hipStream_t s) {
  const HeadPlan pl = plan_head(d, 1);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  HeadParams p{};
  const int rc = head_params(d, pl, p);
  if (rc) return rc;
  p.src = dy; p.wp = wp; p.bias = nullptr; p.out = dx;
  p.srcBS = (long)d->Cout * pl.SL; p.outPS = (long)d->H * d->W; p.outBS = (long)d->Cin * p.outPS;
  p.act = MPA_ACT_NONE; p.slope = 0.f;
  return head_launch(pl, p, d->B, s);
}

int mpa_conv_tall_fwd(const mpa_conv_desc* d, const float* x, const float* wp, const float* bias, float* y, int act, float slope,
                      hipStream_t s) {
  const HeadPlan pl = plan_tall(d, 2);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  HeadParams p{};
  const int rc = head_params(d, pl, p);
  if (rc) return rc;
  p.src = x; p.wp = wp; p.bias = bias; p.out = y;
  p.srcBS = (long)d->Cin * pl.SL; p.outBS = (long)d->Cout * pl.P; p.outPS = 0;
  p.act = act; p.slope = slope;
  return head_launch(pl, p, d->B, s);
}

int mpa_conv_tall_bwd_data(const mpa_conv_desc* d, const float* dy, const float* wp, float* dx, hipStream_t s) {
  const HeadPlan pl = plan_tall(d, 3);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  HeadParams p{};
  const int rc = head_params(d, pl, p);
  if (rc) return rc;
  p.src = dy; p.wp = wp; p.bias = nullptr; p.out = dx;
  p.srcBS = (long)d->Cout * pl.SL; p.outBS = (long)d->Cin * pl.P; p.outPS = 0;
  p.act = MPA_ACT_NONE; p.slope = 0.f;
  return head_launch(pl, p, d->B, s);
}

int64_t mpa_conv_head_wgrad_workspace(const mpa_conv_desc* d) {
  const HeadWgPlan pl = plan_head_wgrad(d);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)pl.S * d->Cout * (d->Cin * 9 + 1) * 4;
}

int mpa_conv_head_bwd_weight(const mpa_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* workspace,
                             int64_t workspace_bytes, hipStream_t s) {
  const HeadWgPlan pl = plan_head_wgrad(d);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  const int64_t need = (int64_t)pl.S * d->Cout * (d->Cin * 9 + 1) * 4;
  if (!workspace || workspace_bytes < need) return MPA_ERR_WORKSPACE;
  HeadWgParams p{};
  p.x = x; p.dy = dy; p.ws = (float*)workspace;
  p.B = d->B; p.Cin = d->Cin; p.Cout = d->Cout; p.H = d->H; p.W = d->W; p.OW = d->W / 3; p.P = pl.P; p.HW = d->H * d->W; p.S = pl.S;
  p.NCS = pl.NCS; p.SEG = pl.SEG; p.KS = pl.SEG / 4; p.NRB = pl.NRB; p.RB = pl.RB;
  p.XPu = pl.XPu; p.DPu = pl.DPu; p.XUs = pl.XUs; p.DUs = pl.DUs;
  p.items = pl.items; p.itemsPer = pl.itemsPer;
  p.xmagic = (int)(((1L << 20) + pl.XPu - 1) / pl.XPu);
  p.dmagic = (int)(((1L << 20) + pl.DPu - 1) / pl.DPu);
  for (long u = 0; u < std::max(pl.XUs, pl.DUs); ++u) {       // the multiply-shift divisions are exact over the units a launch uses
    if (u < pl.XUs && (int)((u * p.xmagic) >> 20) != (int)(u / pl.XPu)) return MPA_ERR_UNSUPPORTED;
    if (u < pl.DUs && (int)((u * p.dmagic) >> 20) != (int)(u / pl.DPu)) return MPA_ERR_UNSUPPORTED;
  }
  p.dbg = mpa_diag().dbg_head;
  int rc;
  switch (pl.MT) {
    case 1: rc = head_wgrad_launch<1>(pl, p, s); break;
    case 2: rc = head_wgrad_launch<2>(pl, p, s); break;
    case 3: rc = head_wgrad_launch<3>(pl, p, s); break;
    case 4: rc = head_wgrad_launch<4>(pl, p, s); break;
    case 5: rc = head_wgrad_launch<5>(pl, p, s); break;
    default: return MPA_ERR_UNSUPPORTED;
  }
  if (rc) return rc;
  return mpa_conv_reduce_partials((const float*)workspace, dw, db, d->Cout, d->Cin * 9, d->Cin * 9 + 1, pl.S, s);
}
