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
// (round 3: this file holds the forward / backward-data kernel, the filter packers and their C ABI; the planners live in
// conv_plan.h, the staging helpers in conv_stage.h, the backward-weight kernels in conv_wgrad.hip / conv_wgrad15.hip)
#include "mpa_common.h"
#define MPA_COMMON_CDIV 1
#include "conv_plan.h"
#include "conv_internal.h"
#include "conv_fwd_params.h"

namespace {

// the kernel instantiations live in conv_fwd_nb12.hip / conv_fwd_nb45.hip / conv_fwd_nb36.hip (conv_fwd_kernel.h)
int launch_fwd(const FwdPlan& pl, const ConvFwdParams& p, hipStream_t s) {
  const MpaFwdLaunch L{pl.NB, pl.PB, pl.KWS, pl.coTiles, pl.nChunks, pl.lds_bytes};
  switch (pl.NB) {
    case 1: case 2: return mpa_conv_fwd_launch_nb12(L, p, s);
    case 4: case 5: return mpa_conv_fwd_launch_nb45(L, p, s);
    case 3: case 6: return mpa_conv_fwd_launch_nb36(L, p, s);
    default: return MPA_ERR_UNSUPPORTED;
  }
}

// ------------------------------------------------------------------------------------------------ filter packing
// packed[cot][chunk][dy][dx][ck][COTP]
//   mode 0 (forward)      : value = w[co][ci][dy][dx]
//   mode 1 (backward-data): the derived conv has Cin' = Cout, Cout' = Cin (stride 1) or kw*Cin (stride == kernel
//                           along W), value = w[ci'][co' % Cin][kh-1-dy][dxsel]
struct PackDims {                  // one packed bank: dims of the conv that will consume it
  int xphase;                      // strided-W backward (dx taken from co' / Cin)
  int yphase;                      // V > 1: cout'' = c*V + v with the (flipped, for backward-data) filter shifted down by v rows
  int CinP, CoutP, kh, kw;
  int CK, nChunks, COT, COTP, coTiles;
  int KWP;                         // > 0: tap-vector layout packed[cot][chunk][dy][ck][COT][KWP] (see fwd_kw_special)
  int co_off;                      // first source channel of the bank's output rows (cout remainder fold: C0)
  long total;
};

struct PackParams {
  const float* w;
  float* wp;
  int Cout_w, Cin_w, kh_w, kw_w;   // original filter dims
  int mode;                        // 0 forward, 1 backward-data (flipped + transposed), 2 forward with the cout remainder fold
  PackDims a, b;                   // b.total > 0: second bank behind the first (cout remainder fold: main channels, fold rows)
  long total;
  // head conv2 (conv_head.hip): packed[chunk][(kq * hNT + tap) * hMTT + mtile][lane 64], hAUw words per chunk;
  // lane (i, kk) of m-tile mt holds A[16 mt + i][chunk * CK + 4 kq + kk][tap].  head 1: forward (row = cout, tap = 3 dy + dx),
  // head 2: backward-data (row = 3 cin + dx, contraction channel = cout, tap = dy)
  int head, hNT, hMTT, hCK, hNG;     // head 3 / 4: tall (kh,1) filters forward / backward-data, hNG tap groups of hNT (plan_tall)
  long hAUw;
};

__device__ __forceinline__ void conv_pack_head_range(const PackParams& p, long first, long step) {
  const long used = (long)(p.hCK / 4) * p.hNT * p.hMTT * 64;
  for (long i = first; i < p.total; i += step) {
    const long chunk = i / p.hAUw;
    long r = i - chunk * p.hAUw;
    float v = 0.f;
    if (r < used) {
      const int lane = (int)(r & 63); r >>= 6;
      const int mt = (int)(r % p.hMTT); r /= p.hMTT;
      const int tap = (int)(r % p.hNT);
      const int kq = (int)(r / p.hNT);
      const int m = 16 * mt + (lane & 15);
      if (p.head >= 3) {               // chunk = (channel chunk, tap group); one column of kh taps
        const int cq = (int)(chunk / p.hNG), tg = (int)(chunk - (long)cq * p.hNG);
        const int k = cq * p.hCK + 4 * kq + (lane >> 4), dy = tg * p.hNT + tap;
        if (p.head == 3) { if (m < p.Cout_w && k < p.Cin_w) v = p.w[((long)m * p.Cin_w + k) * p.kh_w + dy]; }
        else if (m < p.Cin_w && k < p.Cout_w) v = p.w[((long)k * p.Cin_w + m) * p.kh_w + dy];
        p.wp[i] = v;
        continue;
      }
      const int k = (int)chunk * p.hCK + 4 * kq + (lane >> 4);
      if (p.head == 1) {
        if (m < p.Cout_w && k < p.Cin_w) v = p.w[((long)m * p.Cin_w + k) * 9 + tap];
      } else {
        const int ci = m / 3, dx = m - 3 * ci;
        if (ci < p.Cin_w && k < p.Cout_w) v = p.w[(((long)k * p.Cin_w + ci) * 3 + tap) * 3 + dx];
      }
    }
    p.wp[i] = v;
  }
}

__device__ __forceinline__ float conv_pack_value(const PackParams& p, const PackDims& q, long r) {
  int col, ck, dx;
  if (q.KWP > 0) {
    int pos = (int)(r % q.KWP); r /= q.KWP;
    col = (int)(r % q.COTP); r /= q.COTP;
    ck = (int)(r % q.CK); r /= q.CK;
    // 16-tap rows: chunk g of the row sits at position (g + (col&15)>>2) & 3 -- undo the rotation to find the tap
    if (q.KWP == 16) pos = ((((pos >> 2) - ((col & 15) >> 2)) & 3) << 2) | (pos & 3);
    dx = pos;
  } else {
    col = (int)(r % q.COTP); r /= q.COTP;
    ck = (int)(r % q.CK); r /= q.CK;
    dx = (int)(r % q.kw); r /= q.kw;
  }
  const int dy = (int)(r % q.kh); r /= q.kh;
  const int chunk = (int)(r % q.nChunks); r /= q.nChunks;
  const int cot = (int)r;
  const int co = cot * q.COT + col, ci = chunk * q.CK + ck;
  float v = 0.f;
  if (col < q.COT && co < q.CoutP && ci < q.CinP && dx < q.kw) {
    if (p.mode != 1) {                 // forward: w[co][ci][dy][dx]; fold rows: channel co_off + co / V, filter shifted down by co % V
      if (q.yphase > 1) {
        const int cc = co / q.yphase, dyo = dy - (co - cc * q.yphase);
        if (dyo >= 0 && dyo < p.kh_w) v = p.w[(((long)(q.co_off + cc) * p.Cin_w + ci) * p.kh_w + dyo) * p.kw_w + dx];
      } else {
        v = p.w[(((long)(q.co_off + co) * p.Cin_w + ci) * p.kh_w + dy) * p.kw_w + dx];
      }
    } else if (q.yphase > 1) {
      const int cc = co / q.yphase, dyo = dy - (co - cc * q.yphase);
      if (dyo >= 0 && dyo < p.kh_w)
        v = p.w[(((long)ci * p.Cin_w + q.co_off + cc) * p.kh_w + (p.kh_w - 1 - dyo)) * p.kw_w + (p.kw_w - 1 - dx)];
    } else if (!q.xphase) {
      v = p.w[(((long)ci * p.Cin_w + q.co_off + co) * p.kh_w + (p.kh_w - 1 - dy)) * p.kw_w + (p.kw_w - 1 - dx)];
    } else if (q.xphase == 2) {          // stride == kernel both ways: cout' = (cin*kh + v)*kw + q, a 1x1 convolution
      const int nph = p.kh_w * p.kw_w, cc = co / nph, ph = co - cc * nph;
      v = p.w[((long)ci * p.Cin_w + cc) * nph + ph];
    } else {
      const int cc = co / p.kw_w, ph = co - cc * p.kw_w;    // cout' = cin*kw + dx phase (see the epilogue)
      v = p.w[(((long)ci * p.Cin_w + cc) * p.kh_w + (p.kh_w - 1 - dy)) * p.kw_w + ph];
    }
  }
  return v;
}

__device__ __forceinline__ void conv_pack_range(const PackParams& p, long first, long step) {
  if (p.head) { conv_pack_head_range(p, first, step); return; }
  for (long i = first; i < p.total; i += step)
    p.wp[i] = i < p.a.total ? conv_pack_value(p, p.a, i) : conv_pack_value(p, p.b, i - p.a.total);
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


}  // namespace

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

static int pack_params(const mpa_conv_desc* d, int mode, const float* w, float* w_packed, PackParams& p);

int64_t mpa_conv2d_packed_floats(const mpa_conv_desc* d, int mode) {
  if (!d) return MPA_ERR_ARG;
  PackParams p{};
  const int rc = pack_params(d, mode, nullptr, nullptr, p);
  return rc ? rc : (int64_t)p.total;
}

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

static long pack_dims(PackDims& q, const FwdPlan& pl, int CinP, int CoutP, int kh, int kw, int xphase, int yphase, int co_off) {
  q.xphase = xphase; q.yphase = yphase; q.CinP = CinP; q.CoutP = CoutP; q.kh = kh; q.kw = kw;
  q.CK = pl.CK; q.nChunks = pl.nChunks; q.COT = pl.COT; q.COTP = pl.COTP; q.coTiles = pl.coTiles;
  q.KWP = pl.KWS ? pl.KWP : 0;
  q.co_off = co_off;
  q.total = (long)pl.coTiles * pl.nChunks * kh * (pl.KWS ? pl.KWP : kw) * pl.CK * pl.COTP;
  return q.total;
}

// the two launches of a layer with the cout remainder fold (plan_fold): `main` = channels [0, C0), `fold` = V * R rows.
// Forward (mode 2): the layer itself; backward-data (mode 1): the derived convolution (Cin' = Cout, Cout' = Cin).
struct FoldLaunch {
  bool ok;
  FoldPlan f;
  int Cin, H, W, kh, kw, ph, pw;    // the convolution the two launches split (forward: d's; backward-data: the derived one)
  FwdPlan main, fold;
};

static FoldLaunch fold_launch(const mpa_conv_desc* d, int mode) {
  FoldLaunch fl{};
  fl.ok = false;
  if (mode == 2) {
    fl.f = plan_fold(d->Cout, d->kh, d->kw, d->sh, d->sw, d->H);
    fl.Cin = d->Cin; fl.H = d->H; fl.W = d->W; fl.kh = d->kh; fl.kw = d->kw; fl.ph = d->ph; fl.pw = d->pw;
  } else if (mode == 1) {
    const BwdDataGeom g = bwd_data_geom(d);
    if (!g.ok || g.xphase || g.yphase > 1) return fl;
    fl.f = plan_fold(g.Cout, g.kh, g.kw, 1, 1, g.H);
    fl.Cin = g.Cin; fl.H = g.H; fl.W = g.W; fl.kh = g.kh; fl.kw = g.kw; fl.ph = g.ph; fl.pw = g.pw;
  } else {
    return fl;
  }
  if (!fl.f.ok) return fl;
  const int V = fl.f.V;
  const int OH = fl.H + 2 * fl.ph - fl.kh + 1;
  if (OH < 1) return fl;
  fl.main = plan_fwd(d->B, fl.Cin, fl.H, fl.W, fl.f.C0, fl.kh, fl.kw, 1, 1, fl.ph, fl.pw);
  fl.fold = plan_fwd(d->B, fl.Cin, fl.H + V - 1, fl.W, V * fl.f.R, fl.kh + V - 1, fl.kw, V, 1, fl.ph, fl.pw, false, true, 1);
  fl.ok = fl.main.ok && fl.fold.ok;
  return fl;
}

static int pack_params(const mpa_conv_desc* d, int mode, const float* w, float* w_packed, PackParams& p) {
  p.w = w; p.wp = w_packed;
  p.Cout_w = d->Cout; p.Cin_w = d->Cin; p.kh_w = d->kh; p.kw_w = d->kw;
  p.mode = mode;
  if (mode < 0 || mode > 2) return MPA_ERR_ARG;
  if (mode == 0 || mode == 1) {
    const HeadPlan hp = plan_head(d, mode);
    if (hp.ok) {
      p.head = mode + 1; p.hNT = hp.NT; p.hMTT = hp.MTT; p.hAUw = hp.AUw; p.hCK = hp.CK; p.hNG = 1;
      p.total = (long)hp.nChunks * hp.AUw;
      return MPA_OK;
    }
  }
  if (mode == 0 || mode == 1) {
    int NG = 1;
    const HeadPlan tp = plan_tall(d, mode + 2, &NG);
    if (tp.ok) {
      p.head = mode + 3; p.hNT = tp.NT; p.hMTT = tp.MTT; p.hAUw = tp.AUw; p.hCK = tp.CK; p.hNG = NG;
      p.total = (long)tp.nChunks * tp.AUw;
      return MPA_OK;
    }
  }
  if (mode != 0) {
    const FoldLaunch fl = fold_launch(d, mode);
    if (fl.ok) {
      pack_dims(p.a, fl.main, fl.Cin, fl.f.C0, fl.kh, fl.kw, 0, 1, 0);
      pack_dims(p.b, fl.fold, fl.Cin, fl.f.V * fl.f.R, fl.kh + fl.f.V - 1, fl.kw, 0, fl.f.V, fl.f.C0);
      p.total = p.a.total + p.b.total;
      return MPA_OK;
    }
    if (mode == 2) return MPA_ERR_UNSUPPORTED;
  }
  FwdPlan pl;
  if (mode == 0) {
    pl = plan_fwd(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
    if (!pl.ok) return MPA_ERR_UNSUPPORTED;
    pack_dims(p.a, pl, d->Cin, d->Cout, d->kh, d->kw, 0, 1, 0);
  } else {
    BwdDataGeom g = bwd_data_geom(d);
    if (!g.ok) return MPA_ERR_UNSUPPORTED;
    pl = plan_bwd_data(d, g);
    if (!pl.ok) return MPA_ERR_UNSUPPORTED;
    pack_dims(p.a, pl, g.Cin, g.Cout, g.kh, g.kw, g.xyphase ? 2 : (g.xphase ? 1 : 0), g.yphase, 0);
  }
  p.b.total = 0;
  p.total = p.a.total;
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
  p.dbg = mpa_diag().dbg_fwd;
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

// the two launches of fold_launch(): src (x, or dy for backward-data), packed = [main bank][fold bank], dst with Ctot channels
static int conv_folded_impl(const mpa_conv_desc* d, const FoldLaunch& fl, const float* src, const float* packed, const float* bias,
                            float* dst, int act, float slope, hipStream_t s) {
  const int OH = fl.H + 2 * fl.ph - fl.kh + 1, OW = fl.W + 2 * fl.pw - fl.kw + 1, V = fl.f.V;
  const int Ctot = fl.f.C0 + fl.f.R;
  const long outCS = (long)OH * OW, outBS = (long)Ctot * outCS;
  PackDims a{};
  const long a_total = pack_dims(a, fl.main, fl.Cin, fl.f.C0, fl.kh, fl.kw, 0, 1, 0);
  int rc = conv_fwd_impl(d->B, fl.Cin, fl.H, fl.W, fl.f.C0, fl.kh, fl.kw, 1, 1, fl.ph, fl.pw, src, packed, bias, dst, act, slope,
                         outBS, outCS, OW, 1, fl.f.C0, s);
  if (rc) return rc;
  return conv_fwd_impl(d->B, fl.Cin, fl.H, fl.W, V * fl.f.R, fl.kh + V - 1, fl.kw, V, 1, fl.ph, fl.pw, src, packed + a_total,
                       bias ? bias + fl.f.C0 : nullptr, dst + (long)fl.f.C0 * outCS, act, slope, outBS, outCS, OW, 1, fl.f.R, s,
                       false, fl.H + V - 1, V, OH);
}

int mpa_conv2d_fold_supported(const mpa_conv_desc* d) {
  if (!d) return 0;
  return fold_launch(d, 2).ok ? 1 : 0;
}

int mpa_conv2d_fwd_folded(const mpa_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* y, int act,
                          float slope, void* stream) {
  if (!d || !x || !w_packed || !y || d->B <= 0) return MPA_ERR_ARG;
  const FoldLaunch fl = fold_launch(d, 2);
  if (!fl.ok) return MPA_ERR_UNSUPPORTED;
  return conv_folded_impl(d, fl, x, w_packed, bias, y, act, slope, (hipStream_t)stream);
}

int mpa_conv2d_fwd(const mpa_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                   int act, float slope, void* stream) {
  if (!d || !x || !w_packed || !y || d->B <= 0) return MPA_ERR_ARG;
  const int OH = (d->H + 2 * d->ph - d->kh) / d->sh + 1, OW = (d->W + 2 * d->pw - d->kw) / d->sw + 1;
  if (OH <= 0 || OW <= 0) return MPA_ERR_ARG;
  if (plan_head(d, 0).ok) return mpa_conv_head_fwd(d, x, w_packed, bias, y, act, slope, (hipStream_t)stream);
  if (plan_tall(d, 2).ok) return mpa_conv_tall_fwd(d, x, w_packed, bias, y, act, slope, (hipStream_t)stream);
  return conv_fwd_impl(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw, x, w_packed, bias,
                       y, act, slope, (long)d->Cout * OH * OW, (long)OH * OW, OW, 1, d->Cout, (hipStream_t)stream);
}

int64_t mpa_conv2d_fwd_stats_rows(const mpa_conv_desc* d) {
  if (!d) return MPA_ERR_ARG;
  if (plan_head(d, 0).ok || plan_tall(d, 2).ok) return MPA_ERR_UNSUPPORTED;
  FwdPlan pl = plan_fwd(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
  if (!pl.ok) return MPA_ERR_UNSUPPORTED;
  return (int64_t)d->B * pl.tilesY * pl.tilesX;
}

int mpa_conv2d_fwd_stats(const mpa_conv_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                         float* partials, void* stream) {
  if (!d || !x || !w_packed || !y || !partials || d->B <= 0) return MPA_ERR_ARG;
  const int OH = (d->H + 2 * d->ph - d->kh) / d->sh + 1, OW = (d->W + 2 * d->pw - d->kw) / d->sw + 1;
  if (OH <= 0 || OW <= 0) return MPA_ERR_ARG;
  if (plan_head(d, 0).ok || plan_tall(d, 2).ok) return MPA_ERR_UNSUPPORTED;      // mode-0 banks of that geometry are in conv_head.hip's layout
  return conv_fwd_impl(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw, x, w_packed, bias,
                       y, MPA_ACT_NONE, 0.f, (long)d->Cout * OH * OW, (long)OH * OW, OW, 1, d->Cout, (hipStream_t)stream,
                       false, 0, 1, 0, partials);
}

int mpa_conv2d_bwd_data(const mpa_conv_desc* d, const float* dy, const float* w_packed, float* dx, void* stream) {
  if (!d || !dy || !w_packed || !dx || d->B <= 0) return MPA_ERR_ARG;
  if (plan_head(d, 1).ok) return mpa_conv_head_bwd_data(d, dy, w_packed, dx, (hipStream_t)stream);
  if (plan_tall(d, 3).ok) return mpa_conv_tall_bwd_data(d, dy, w_packed, dx, (hipStream_t)stream);
  {
    const FoldLaunch fl = fold_launch(d, 1);
    if (fl.ok) return conv_folded_impl(d, fl, dy, w_packed, nullptr, dx, MPA_ACT_NONE, 0.f, (hipStream_t)stream);
  }
  BwdDataGeom g = bwd_data_geom(d);
  if (!g.ok) return MPA_ERR_UNSUPPORTED;
  const long inBS = (long)d->Cin * d->H * d->W, inCS = (long)d->H * d->W;
  if (g.yphase > 1) {
    // V output rows per cout block: derived conv with vertical stride V; row oy*V + v of channel cout''/V
    return conv_fwd_impl(d->B, g.Cin, g.H, g.W, g.Cout, g.kh, g.kw, g.sh, 1, g.ph, g.pw, dy, w_packed, nullptr, dx,
                         MPA_ACT_NONE, 0.f, inBS, inCS, d->W, 1, d->Cin, (hipStream_t)stream, true, g.Hplan, g.yphase,
                         d->H);
  }
  if (g.xyphase) {
    const int OH = (d->H - d->kh) / d->sh + 1, OW = (d->W - d->kw) / d->sw + 1;
    if (OH * d->sh != d->H || OW * d->sw != d->W)          // rows / columns behind the last window: no gradient
      if (mpa_zero_async(dx, sizeof(float) * (size_t)d->B * (size_t)inBS, (hipStream_t)stream) != MPA_OK) return MPA_ERR_LAUNCH;
    return conv_fwd_impl(d->B, g.Cin, g.H, g.W, g.Cout, 1, 1, 1, 1, 0, 0, dy, w_packed, nullptr, dx, MPA_ACT_NONE, 0.f, inBS, inCS,
                         d->W, d->sw, d->Cin, (hipStream_t)stream, true, 0, d->sh, d->H);
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
  if (mode == 0 || mode == 1) {
    int NG = 1;
    const HeadPlan tp = plan_tall(d, mode + 2, &NG);
    if (tp.ok) {
      snprintf(buf, buflen, "tall_gemm<%d,%d,%d> rows=%d tiles=%d K=%d taps=%dx%d chunks=%d px/wg=%d tiles/img=%d XS=%d lds=%zuB", tp.MT,
               tp.WM, tp.WN, tp.Mrows, tp.MTT, tp.K, NG, tp.NT, tp.nChunks, tp.PXT, tp.tilesP, tp.XS, tp.lds_bytes);
      return MPA_OK;
    }
  }
  if (mode == 0 || mode == 1) {
    const HeadPlan hp = plan_head(d, mode);
    if (hp.ok) {
      snprintf(buf, buflen, "head_gemm<%d,%d,%d> rows=%d tiles=%d K=%d CK=%d chunks=%d taps=%d px/wg=%d tiles/img=%d XS=%d lds=%zuB",
               hp.MT, hp.WM, hp.WN, hp.Mrows, hp.MTT, hp.K, hp.CK, hp.nChunks, hp.NT, hp.PXT, hp.tilesP, hp.XS, hp.lds_bytes);
      return MPA_OK;
    }
  }
  if (mode == 2) {
    const HeadWgPlan hw = plan_head_wgrad(d);
    if (hw.ok) {
      snprintf(buf, buflen, "head_wgrad<%d> coGroups=%d chGroups=%d S=%d items/slice=%ld segments=%dx%dpx rowblocks=%dx%d lds=%zuB",
               hw.MT, hw.coGroups, hw.chGroups, hw.S, hw.itemsPer, hw.NCS, hw.SEG, hw.NRB, hw.RB, hw.lds_bytes);
      return MPA_OK;
    }
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
  if (mode == 1 || mode == 3) {            // 3: the forward pass with the cout remainder fold (pack mode 2)
    const FoldLaunch fl = fold_launch(d, mode == 3 ? 2 : 1);
    if (fl.ok) {
      snprintf(buf, buflen, "fold %d + %dx%d rows: main fwd<%d,%d> COT=%d coTiles=%d tile=%dx%d tiles=%dx%d kwvec=%d | fold fwd<%d,%d> kh=%d stride=%d tile=%dx%d tiles=%dx%d kwvec=%d lds=%zuB",
               fl.f.C0, fl.f.R, fl.f.V, fl.main.NB, fl.main.PB, fl.main.COT, fl.main.coTiles, fl.main.TH, fl.main.TW, fl.main.tilesY,
               fl.main.tilesX, fl.main.KWS, fl.fold.NB, fl.fold.PB, fl.kh + fl.f.V - 1, fl.f.V, fl.fold.TH, fl.fold.TW, fl.fold.tilesY,
               fl.fold.tilesX, fl.fold.KWS, std::max(fl.main.lds_bytes, fl.fold.lds_bytes));
      return MPA_OK;
    }
    if (mode == 3) return MPA_ERR_UNSUPPORTED;
  }
  FwdPlan f;
  if (mode == 0) f = plan_fwd(d->B, d->Cin, d->H, d->W, d->Cout, d->kh, d->kw, d->sh, d->sw, d->ph, d->pw);
  else {
    BwdDataGeom g = bwd_data_geom(d);
    if (!g.ok) return MPA_ERR_UNSUPPORTED;
    f = plan_bwd_data(d, g);
  }
  if (!f.ok) return MPA_ERR_UNSUPPORTED;
  snprintf(buf, buflen, "fwd<%d,%d> COT=%d coTiles=%d CK=%d chunks=%d tile=%dx%d tiles=%dx%d halo=%dx%d LW=%d quad=%d kwvec=%d ksplit=%d lds=%zuB cost=%.4g",
           f.NB, f.PB, f.COT, f.coTiles, f.CK, f.nChunks, f.TH, f.TW, f.tilesY, f.tilesX, f.IH, f.IW, f.LW, f.quad, f.KWS, f.KS, f.lds_bytes,
           f.cost);
  return MPA_OK;
}

}  // extern "C"
