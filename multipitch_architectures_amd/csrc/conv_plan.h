// Host-side planners of the fp32 convolution kernels (tile shapes, LDS pitches, channel chunks, slice counts) -- plain
// C++, no HIP types: what was the first third of conv.hip until round 3.  Every conv*.hip includes this header (the
// functions have internal linkage); tests/test_cpu_plan.py compiles it with g++ alone and checks plans on the CPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/mpa.h"
#include "mpa_diag.h"

namespace {

inline int64_t mpa_cdiv_plan(int64_t a, int64_t b) { return (a + b - 1) / b; }
#ifndef MPA_COMMON_CDIV
#define mpa_cdiv mpa_cdiv_plan
#define MPA_PLAN_OWN_CDIV 1
#endif

// ------------------------------------------------------------------------------------------------ planning
constexpr long FWD_LDS_BUDGET = 78 * 1024;   // two workgroups per CU (160 KiB LDS)
constexpr int EDGE_MAXF = 3;                 // straddling quads (rows) per thread that edge_fix_* can carry (see there)

struct FwdPlan {
  int NB, PB, TH, TW, tilesY, tilesX, CK, nChunks, IH, IW, LW, CHP, COT, COTP, coTiles, OH, OW, quad;
  size_t lds_bytes;
  bool ok;
  int KWS, KWP;   // tap-vector filter layout (kw 15 / 9, PB >= 4): kernel specialised on kw, slab rows of KWP taps
  int KS;         // input-channel split: blockIdx.z owns nChunks/KS chunks and adds its partial sums atomically
  double cost = 0.0;   // the cost model's value for this plan (describe_plan)
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
  const MpaDiag& diag = mpa_diag();          // diagnostics (mpa_diag.h): fwd_nb, fwd_pb restrict the search
  const int fNB = diag.fwd_nb, fPB = diag.fwd_pb;
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
          if ((diag.fwd_th && TH != diag.fwd_th) || (diag.fwd_tw && TW != diag.fwd_tw)) continue;   // diagnostics
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
            const double halo_w = (NB == 1 && diag.fwd_halo_nb1 >= 0.0) ? diag.fwd_halo_nb1 : 0.05;
            const double per_block = (double)P * NB * (1.0 + halo_w * IH * IW / P + opnd) * (1.0 + 0.02 * lwi) *
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
            const int ks_max = diag.fwd_ks_max, ks_force = diag.fwd_ks_force;   // diagnostics
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
              best.cost = cost;
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

struct BwdDataGeom {
  bool ok, xphase;
  bool xyphase;                          // stride == kernel in both directions: phases (v, q) of a 1x1 convolution
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
  } else if (d->sh == d->kh && d->sw == d->kw && d->ph == 0 && d->pw == 0 && d->kh * d->kw <= 16) {
    // non-overlapping windows in both directions (basic_cnn's conv2: 3x3, stride (3,3), basic_cnns.py:39): a 1x1 convolution
    // to kh*kw*Cin phase channels, cout' = (cin*kh + v)*kw + q stored at (oy*kh + v, ox*kw + q); input rows / columns past
    // the last full window get no gradient (the caller zero-fills dx when the windows do not tile the plane)
    g.ok = true; g.xphase = true; g.xyphase = true;
    g.Cout = d->kh * d->kw * d->Cin; g.kh = 1; g.kw = 1; g.ph = 0; g.pw = 0;
  } else {
    g.ok = false;
  }
  return g;
}

inline FwdPlan plan_bwd_data(const mpa_conv_desc* d, const BwdDataGeom& g) {
  return plan_fwd(d->B, g.Cin, g.Hplan, g.W, g.Cout, g.kh, g.kw, g.sh, 1, g.ph, g.pw, true, g.xphase || g.yphase > 1,
                  g.xphase ? d->sw : 1);
}

// ------------------------------------------------------------------------------------------------ cout remainder fold
// 15x15 stride-1 layers whose output channels are not a multiple of 16 (the CNN families: 70 = 64 + 6, 20 = 16 + 4,
// 40 = 32 + 8, 100 = 96 + 4) pad the last MFMA tile: 70 couts run as 80.  Forward and backward-data (whose derived conv
// has Cout' = Cin) then take two launches into the same tensor: channels [0, C0) as an ordinary convolution and the R
// remaining channels as V * R rows of one 16-row tile -- V vertically adjacent output rows per channel, a (kh + V - 1)-row
// filter that is the original shifted down by v rows, vertical stride V (the mapping bwd_data_yphase uses for the 6-channel
// input layer).  12 of 16 rows busy for 16/15 of the taps instead of 6 of 16.  The backward-weight has its own tap fold
// (conv_wgrad15.hip).
struct FoldPlan {
  bool ok;
  int C0, R, V;      // main channels, remainder channels, output rows per remainder channel and launch row
};

inline FoldPlan plan_fold(int Cout_out, int kh, int kw, int sh, int sw, int H) {
  FoldPlan f{};
  f.ok = false;
  if (kh != 15 || kw != 15 || sh != 1 || sw != 1 || mpa_diag().fold_off) return f;
  f.R = Cout_out % 16;
  f.C0 = Cout_out - f.R;
  if (f.R < 1 || f.R > 8 || f.C0 < 16) return f;
  f.V = f.R <= 4 ? 4 : 2;
  if (H < f.V) return f;
  f.ok = true;
  return f;
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
  const MpaDiag& diag = mpa_diag();        // diagnostics / tests: wg_variant restricts the search to one wave tile
  for (int v = 0; v < 5; ++v) {
    if (diag.wg_variant >= 0 && diag.wg_variant != v) continue;
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
    const bool old_norm = diag.wg_costnorm_old;   // diagnostics
    const double pad_eff = old_norm ? ((double)pl.coTiles * pl.COT / d->Cout) * ((double)pl.nTiles * pl.nPerBlock / Ntot)
                                    : (double)pl.coTiles * pl.nTiles / 8.0;
    const int force_txn = diag.wg_txn;   // diagnostics
    // dY-from-global variant: exact tiling of 4-aligned rows, stride 1 (any variant) or the head's stride 3 (<5,6>)
    const bool no_ga = diag.wg_no_ga, force_ga = diag.wg_force_ga;   // diagnostics / tests: MPA_WG_GA = "0" never, "force" whenever feasible
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
  if (totalTiles / S < 8 && !diag.wg_s_old) {
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
  if (diag.wg_s) {      // diagnostics: force the slice count
    const long f = diag.wg_s;
    if (f >= 1 && f <= std::min<long>(totalTiles, 1024)) S = f;
  }
  best.S = (int)S;
  return best;
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
  if (OW % 4 == 0 && !mpa_diag().wg15_lds_dy) {
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
  if (pl.ga && !mpa_diag().wg15_nofold) {
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
  if (totalTiles / S < 8 && !mpa_diag().wg15_s_old) {
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
  if (mpa_diag().wg15_s) {      // diagnostics: force the slice count
    const long f = mpa_diag().wg15_s;
    if (f >= 1 && f <= std::min<long>(totalTiles, 1024)) S = f;
  }
  pl.S = (int)S;
  return pl;
}

// ------------------------------------------------------------------------------------------------ head conv2 (conv_head.hip)
// nn.Conv2d(n0, n1, (3,3), stride (1,3), padding (1,0)) of every model's head (unet_cnns.py:538-543, basic_cnns.py:390-395).
// The column stride equals the filter width, so output pixel p = oy*OW + ox of an image reads the input plane (flattened,
// q = y*W + x) at 3p + (dy-1)*W + dx: linear in p.  All three passes are therefore plain GEMMs whose B operand is a
// shifted, strided view of a contiguous slab -- no tile edges along a row, no 2-D halo:
//   forward        y [co][p]        = sum_{ci,dy,dx} w[co][ci][dy][dx] * x [ci][3p + (dy-1)W + dx]
//   backward-data  dx[ci][3p + dx]  = sum_{co,dy}    w[co][ci][dy][dx] * dy[co][p - (dy-1)OW]
//   backward-weight dw[co][ci][dy][dx] = sum_{b,p}   dy[co][p] * x[ci][3p + (dy-1)W + dx]
// Rows above / below the image are zero pages of the DMA staging.
struct HeadPlan {
  bool ok;
  int mode;            // 0 forward, 1 backward-data
  int MT, WM, NB, WN;  // wave tile: MT 16-row tiles x NB 16-pixel blocks; workgroup = WM x WN waves
  int CK, nChunks;     // contraction channels per staged chunk
  int NT;              // taps per contraction channel (9 / 3)
  int K, Mrows, MTT;   // contraction channels, valid output rows (Cout / 3*Cin), WM*MT
  int P, SL, SN, HALO; // pixels per image; source plane words; source words per pixel; halo words either side
  int XS, PXT, tilesP; // slab channel stride (== 16 mod 32: conflict-free ds_read_b32); pixels per workgroup; tiles per image
  long AUw;            // words of one packed filter chunk (multiple of 256 = one 64-lane 16-byte DMA)
  size_t lds_bytes;
};

inline bool head_geom_ok(const mpa_conv_desc* d) {
  return d->kh == 3 && d->kw == 3 && d->sh == 1 && d->sw == 3 && d->ph == 1 && d->pw == 0 && d->W % 12 == 0 && d->W >= 12 &&
         d->H >= 1 && ((long)d->H * d->W) % 4 == 0 && !mpa_diag().head_off;
}

inline HeadPlan plan_head(const mpa_conv_desc* d, int mode) {
  HeadPlan pl{};
  pl.ok = false;
  if (!head_geom_ok(d)) return pl;
  const int OW = d->W / 3;
  pl.mode = mode;
  pl.P = d->H * OW;
  pl.NB = 5;
  if (mode == 0) {
    pl.K = d->Cin; pl.Mrows = d->Cout; pl.NT = 9; pl.SN = 3; pl.HALO = d->W; pl.SL = d->H * d->W;
    const int tiles = (int)mpa_cdiv(d->Cout, 16);
    if (tiles < 4 || tiles > 14 || d->Cin < 16) return pl;
    if (tiles <= 7) { pl.WM = 1; pl.MT = tiles; }
    else { pl.WM = 2; pl.MT = (int)mpa_cdiv(tiles, 2); }
    pl.WN = 4;
    // one workgroup per 320 pixels and all couts: below ~2 workgroups per CU the generic forward kernel (smaller tiles, cout
    // tiles in the grid) fills the chip better -- measured (scratch/head_time.py, TFLOP/s head / generic): 128 -> 80 at batch 16
    // 65 / 60, batch 32 83 / 84, batch 48 93 / 61, batch 256 113 / 102; 128 -> 200 at batch 16 60 / 66, batch 32 80 / 76.
    // Backward-data and backward-weight win at every batch.
    long min_wgs = 512;
    if (mpa_diag().head_fwd_min_wgs >= 0) min_wgs = mpa_diag().head_fwd_min_wgs;
    if ((long)d->B * mpa_cdiv(pl.P, 4 * pl.NB * 16) < min_wgs) return pl;
  } else {
    pl.K = d->Cout; pl.Mrows = 3 * d->Cin; pl.NT = 3; pl.SN = 1; pl.HALO = OW; pl.SL = d->H * OW;
    const int tiles = (int)mpa_cdiv(3 * d->Cin, 16);
    if (d->Cout < 16) return pl;
    if (tiles >= 8 && tiles <= 14) { pl.WM = 2; pl.MT = (int)mpa_cdiv(tiles, 2); pl.WN = 4; }
    else if (tiles >= 17 && tiles <= 28) { pl.WM = 4; pl.MT = (int)mpa_cdiv(tiles, 4); pl.WN = 2; }
    else return pl;
  }
  pl.MTT = pl.WM * pl.MT;
  pl.PXT = pl.WN * pl.NB * 16;
  pl.tilesP = (int)mpa_cdiv(pl.P, pl.PXT);
  pl.XS = round_mod(pl.SN * pl.PXT + 2 * pl.HALO, 32, 16);
  // channels per chunk: 4 when that makes room for two workgroups per CU (4-wave forward tiles: the second workgroup's
  // MFMAs cover this one's staging burst and barrier: 111 against 101 TFLOP/s), else 8 if it fits, else 4
  auto lds_of = [&](int ck) {
    pl.CK = ck;
    pl.AUw = mpa_cdiv((long)(ck / 4) * pl.NT * pl.MTT * 64, 256) * 256;
    pl.lds_bytes = 2 * ((size_t)mpa_cdiv((long)ck * (pl.XS / 4), 64) * 256 + (size_t)pl.AUw) * 4;
    return pl.lds_bytes;
  };
  int ck = mpa_diag().head_ck;
  if (!ck) ck = (pl.WM * pl.WN == 4 && lds_of(4) <= 78 * 1024) ? 4 : 8;
  if (lds_of(ck) > 150 * 1024) {
    if (ck == 8 && lds_of(4) <= 150 * 1024) ck = 4;
    else return pl;
  }
  lds_of(ck);
  if (pl.CK < 4) return pl;
  pl.nChunks = (int)mpa_cdiv(pl.K, pl.CK);
  pl.ok = true;
  return pl;
}

// Tall filters -- conv3's Conv2d(n1, n2, (75,1)) on patches longer than 75 frames (unet_cnns.py:545-549, basic_cnns.py:397-401;
// training / evaluation on segments, T = 174 in the torchinfo summaries) -- run on the same GEMM kernel (conv_head.hip,
// MODE 2): with one column of taps, output pixel p = oy*W + x reads input word p + dy*W, again linear in p.  The kh taps go
// in groups of NT (25, or 15 above 5 output-row tiles: the filter chunk has to fit LDS next to the slab); a chunk = (4
// channels, tap group), its slab = the group's window [p0 + t0*W, p0 + PXT + (t0 + NT - 1)*W) of the plane.  Backward-data =
// the same with the windows running upwards (p - dy*W) and zero pages above / below the dY plane.  HeadPlan fields as above
// (mode 2 forward, 3 backward-data); NG tap groups.
inline HeadPlan plan_tall(const mpa_conv_desc* d, int mode, int* NGout = nullptr) {
  HeadPlan pl{};
  pl.ok = false;
  const int OH = d->H - d->kh + 1;
  if (d->kw != 1 || d->sh != 1 || d->sw != 1 || d->ph != 0 || d->pw != 0 || d->kh < 30 || OH < 2 || d->W % 4 || d->Cin < 16 ||
      d->Cout < 16 || mpa_diag().tall_off)
    return pl;
  const int rows = mode == 2 ? d->Cout : d->Cin;
  const int tiles = (int)mpa_cdiv(rows, 16);
  if (tiles < 2 || tiles > 14) return pl;
  if (tiles <= 7) { pl.WM = 1; pl.MT = std::max(tiles, 4); }
  else { pl.WM = 2; pl.MT = std::max((int)mpa_cdiv(tiles, 2), 4); }
  pl.MTT = pl.WM * pl.MT;
  pl.NT = pl.MTT <= 5 ? 25 : 15;
  if (d->kh % pl.NT) return pl;
  const int NG = d->kh / pl.NT;
  if (NGout) *NGout = NG;
  pl.mode = mode; pl.WN = 4; pl.NB = 5; pl.SN = 1;
  pl.PXT = pl.WN * pl.NB * 16;
  if (mode == 2) { pl.K = d->Cin; pl.Mrows = d->Cout; pl.P = OH * d->W; pl.SL = d->H * d->W; pl.HALO = 0; }
  else { pl.K = d->Cout; pl.Mrows = d->Cin; pl.P = d->H * d->W; pl.SL = OH * d->W; pl.HALO = (pl.NT - 1) * d->W; }
  if (pl.SL % 4 || pl.P % 4) return pl;
  pl.tilesP = (int)mpa_cdiv(pl.P, pl.PXT);
  pl.XS = round_mod(pl.PXT + (pl.NT - 1) * d->W, 32, 16);
  if (pl.XS > 65535) return pl;
  pl.CK = 4;
  pl.AUw = mpa_cdiv((long)pl.NT * pl.MTT * 64, 256) * 256;
  pl.lds_bytes = 2 * ((size_t)mpa_cdiv((long)pl.CK * (pl.XS / 4), 64) * 256 + (size_t)pl.AUw) * 4;
  if (pl.lds_bytes > 150 * 1024) return pl;
  pl.nChunks = (int)mpa_cdiv(pl.K, pl.CK) * NG;
  pl.ok = true;
  return pl;
}

// backward-weight: a workgroup (4 waves x 16 input channels, MT cout tiles, all 9 taps in 36*MT accumulator registers) owns
// work items (image, column segment of SEG output pixels, block of rows) and walks the rows of an item top to bottom with a
// ring of three input-row segments in LDS: every input row is staged once and serves the three filter rows of the output
// rows around it.  Partial sums per slice, reduced in a fixed order.
struct HeadWgPlan {
  bool ok;
  int MT, coGroups, chGroups, S;   // cout tiles per workgroup (<= 5), cout groups, 64-channel groups, slices
  int P, NCS, SEG, NRB, RB;        // pixels per image; column segments and their width; row blocks and their height
  long items, itemsPer;            // work items (B * NCS * NRB), items per slice
  int XPu, DPu, XUs, DUs;          // 16-byte units: X channel pitch, dY row pitch, one X ring slot, one dY buffer
  size_t lds_bytes;
};

inline HeadWgPlan plan_head_wgrad(const mpa_conv_desc* d) {
  HeadWgPlan pl{};
  pl.ok = false;
  if (!head_geom_ok(d) || d->Cin < 16 || d->Cout < 16) return pl;
  const int OW = d->W / 3;
  pl.P = d->H * OW;
  const int tiles = (int)mpa_cdiv(d->Cout, 16);
  pl.coGroups = (int)mpa_cdiv(tiles, 5);
  pl.MT = (int)mpa_cdiv(tiles, pl.coGroups);
  pl.chGroups = (int)mpa_cdiv(d->Cin, 64);
  for (pl.NCS = 1; pl.NCS <= OW / 4; ++pl.NCS) {
    if (OW % (4 * pl.NCS)) continue;
    pl.SEG = OW / pl.NCS;
    // LDS pitches in 16-byte units, odd: lane j of an operand read sits at j * pitch words, and with pitch / 4 odd the 16
    // lanes spread over 8 bank groups (2-way); an even unit count folds them onto 2 or 4 (measured with the first version,
    // pitch 28 units: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.82)
    pl.XPu = (3 * pl.SEG / 4 + 1) | 1;
    pl.DPu = (pl.SEG / 4 + 1) | 1;
    pl.XUs = (int)(mpa_cdiv(64L * pl.XPu, 64) * 64);
    pl.DUs = (int)(mpa_cdiv(16L * pl.MT * pl.DPu, 64) * 64);
    pl.lds_bytes = (size_t)(3 * pl.XUs + 2 * pl.DUs) * 16;
    // (a wave stages a ring slot in at most 8 and a dY buffer in at most 6 DMA instructions: HEAD_WG_NX / _ND)
    if (pl.lds_bytes <= 150 * 1024 && pl.XUs <= 8 * 256 && pl.DUs <= 6 * 256) break;
  }
  if (pl.NCS > OW / 4) return pl;
  const long groups = (long)pl.coGroups * pl.chGroups;
  const long want = std::max<long>(1, 256 / std::min<long>(groups, 256));       // slices that fill the chip once
  // row blocks: as few as fill the chip evenly -- every block re-reads two halo rows (small batches need the split: 32 images
  // x 2 segments are 64 items for 128 slices)
  {
    double best = -1.0;
    pl.NRB = 1;
    for (int n = 1; n <= std::min(d->H, 16); ++n) {
      const int RB = (int)mpa_cdiv(d->H, n), nrb = (int)mpa_cdiv(d->H, RB);
      if (nrb != n) continue;
      const long items = (long)d->B * pl.NCS * nrb, S0 = std::min<long>(items, want), per = mpa_cdiv(items, S0), S1 = mpa_cdiv(items, per);
      const double eff = (double)items / (double)(per * S1) * std::min(1.0, (double)(S1 * groups) / 256.0) / (1.0 + 2.0 / RB);
      if (eff > best + 1e-3) { best = eff; pl.NRB = n; }
    }
  }
  pl.RB = (int)mpa_cdiv(d->H, pl.NRB);
  pl.NRB = (int)mpa_cdiv(d->H, pl.RB);
  pl.items = (long)d->B * pl.NCS * pl.NRB;
  long S = std::min<long>(pl.items, want);
  if (const long f = mpa_diag().head_wg_s) { if (f >= 1 && f <= pl.items) S = f; }
  pl.itemsPer = mpa_cdiv(pl.items, S);
  pl.S = (int)mpa_cdiv(pl.items, pl.itemsPer);
  pl.ok = true;
  return pl;
}

#ifdef MPA_PLAN_OWN_CDIV
#undef mpa_cdiv
#undef MPA_PLAN_OWN_CDIV
#endif

}  // namespace
