// Patch extraction + augmentation on resident recordings.
// Replaces libdl/data_loaders/hcqt_datasets.py:67-141 (dataset_context.__getitem__) and the time-scaling-free part of
// :199-289 (dataset_context_segm.__getitem__) for a whole batch at once: the recordings stay in HBM as
// (n_harm, T_file, n_bins) fp32, one launch gathers B windows and applies EQ -> noise+abs -> log compression ->
// tuning shift -> transposition in registers.  HBM-bound: 4 B read + 4 B written per output element.
#include "mpa_common.h"
#include <math.h>

namespace {

struct CtxParams {
  int n_harm, n_bins, frames, n_out, seglength, flags, B;
  float compression, noisestd;
  int eq_off[16];
  const uint64_t* src;            // [B] device address of (harmonic 0, first frame of the window, bin 0)
  const int64_t* chan_stride;     // [B] elements between harmonics of that recording
  const uint64_t* tgt;            // [B] device address of the first target row
  const int32_t* aug;             // [B][4] alpha, beta, tune2 (half bins), transp (semitones); nullable = all zero
  const float *n1, *n2, *n3;      // explicit N(0,1)*std draws (tests) or null = counter-based generator
  uint64_t seed;
  float* X;
  float* y;
};

// 32-bit avalanche hash (two multiply-xorshift rounds)
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
// two independent N(0,1) values per (seed, stream, index): Box-Muller (both branches) on two 24-bit uniforms obtained
// from a counter hash; fast intrinsics (~1e-6 abs error) are ample for augmentation noise
__device__ __forceinline__ float2 gauss2(uint64_t seed, uint32_t stream, uint64_t idx) {
  const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
  const uint32_t k = hash32((uint32_t)seed ^ (stream * 0x9E3779B9u) ^ hash32(hi + (uint32_t)(seed >> 32)));
  const uint32_t a = hash32(lo ^ k), b = hash32(lo + 0x632BE5ABu + (k << 1 | 1u));
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);           // (0,1]
  const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);                    // [0,1)
  const float rad = sqrtf(-2.0f * __logf(u1));
  float sn, cs;
  __sincosf(6.2831853f * u2, &sn, &cs);
  return make_float2(rad * cs, rad * sn);
}
__device__ __forceinline__ float gauss(uint64_t seed, uint32_t stream, uint64_t idx) { return gauss2(seed, stream, idx).x; }

// stages EQ -> noise+abs -> compression of one bin (hcqt_datasets.py:83-106); `n` = the (already scaled) noise draw
__device__ __forceinline__ float base_value(const CtxParams& p, float v, float n, int g, float eq_scale, int eq_centre) {
  if (p.flags & MPA_CTX_EQ) {
    const int d = g - eq_centre;
    v = (1.0f - eq_scale * (float)(d * d)) * v;
  }
  if (p.flags & MPA_CTX_NOISE) v = fabsf(v + n);
  if (p.flags & MPA_CTX_LOG) v = logf(1.0f + p.compression * v);
  return v;
}

// CTX_ROWS (patch, harmonic, frame) rows per block.  Phase 1 puts the compressed row (stages EQ..log) into LDS once,
// phase 2 forms the tuning shift / transposition from it -- the +-0.5-bin average needs two neighbours per output and
// the Gaussian generator is the expensive part, so nothing is evaluated twice.
constexpr int CTX_ROWS = 4;
constexpr int CTX_MAXBINS = 256;
__global__ __launch_bounds__(256) void context_rows_kernel(CtxParams p, int nrows) {
  __shared__ float base_s[CTX_ROWS][CTX_MAXBINS];
  const int r = threadIdx.x >> 6, lane = threadIdx.x & 63;            // one wave per row: the row decode is wave-uniform
  const int ridx = blockIdx.x * CTX_ROWS + r;                         // (b*n_harm + h)*frames + t
  if (ridx >= nrows) return;                                          // whole wave; no block-wide barrier below
  const int t = ridx % p.frames;
  const int bh = ridx / p.frames;
  const int h = bh % p.n_harm;
  const int b = bh / p.n_harm;
  const float* row = (const float*)p.src[b] + (long)h * p.chan_stride[b] + (long)t * p.n_bins;
  int alpha = 0, beta = 0, tune2 = 0, transp = 0;
  if (p.aug) {
    alpha = p.aug[b * 4 + 0]; beta = p.aug[b * 4 + 1]; tune2 = p.aug[b * 4 + 2]; transp = p.aug[b * 4 + 3];
  }
  if (!(p.flags & MPA_CTX_TUNE)) tune2 = 0;
  if (!(p.flags & MPA_CTX_TRANSP)) transp = 0;
  const float eq_scale = 2e-6f * (float)alpha;
  const int eq_centre = beta - p.eq_off[h];
  const float* n1row = p.n1 ? p.n1 + (long)ridx * p.n_bins : nullptr;
  float* base = base_s[r];
  for (int g = lane; g < p.n_bins; g += 128) {                        // bins g and g+64 share one Box-Muller pair
    const int g2 = g + 64;
    float2 n = make_float2(0.f, 0.f);
    if (p.flags & MPA_CTX_NOISE) {
      if (n1row) {
        n.x = n1row[g];
        if (g2 < p.n_bins) n.y = n1row[g2];
      } else {
        n = gauss2(p.seed, 0, (uint64_t)ridx * 256 + g);
        n.x *= p.noisestd; n.y *= p.noisestd;
      }
    }
    base[g] = base_value(p, row[g], n.x, g, eq_scale, eq_centre);
    if (g2 < p.n_bins) base[g2] = base_value(p, row[g2], n.y, g2, eq_scale, eq_centre);
  }
  __builtin_amdgcn_wave_barrier();                                    // LDS row is private to this wave
  __builtin_amdgcn_s_waitcnt(0xc07f);                                 // lgkmcnt(0): the row is written
  float* out = p.X + (long)ridx * p.n_bins;
  for (int f = lane; f < p.n_bins; f += 64) {
    float v;
    const int g = f - 3 * transp;                     // bin of the tuned row that the +-semitone roll moves to f (:128)
    if (g < 0 || g >= p.n_bins) {                     // exposed by the roll: |N(0,1e-4)| (:131-135)
      const int j = transp > 0 ? f : f - (p.n_bins + 3 * transp);
      const float n = p.n3 ? p.n3[(long)ridx * 15 + j] : 1e-4f * gauss(p.seed, 2, (uint64_t)ridx * 16 + j);
      v = fabsf(n);
    } else if ((tune2 > 0 && g == 0) || (tune2 < 0 && g == p.n_bins - 1)) {   // edge exposed by the tuning shift (:121-124)
      const float n = p.n2 ? p.n2[ridx] : 1e-4f * gauss(p.seed, 1, (uint64_t)ridx);
      v = fabsf(n);
    } else if (tune2 == 1) {                          // +0.5 bin: mean with the lower neighbour (:114-115)
      v = (base[g - 1] + base[g]) / 2;
    } else if (tune2 == -1) {                         // -0.5 bin (:117-118)
      v = (base[g] + base[g + 1]) / 2;
    } else {                                          // 0 or +-1 bin roll (:120)
      v = base[g - tune2 / 2];
    }
    out[f] = v;
  }
}

// targets: (B, 1, seglength, n_out); rolled by the transposition with zero fill (circular for 12 pitch classes, :129-137)
__global__ __launch_bounds__(256) void context_targets_kernel(CtxParams p) {
  const long n = (long)p.B * p.seglength * p.n_out;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const int k = (int)(i % p.n_out);
    const int s = (int)((i / p.n_out) % p.seglength);
    const int b = (int)(i / ((long)p.n_out * p.seglength));
    const int transp = (p.aug && (p.flags & MPA_CTX_TRANSP)) ? p.aug[b * 4 + 3] : 0;
    const float* trow = (const float*)p.tgt[b] + (long)s * p.n_out;
    int src = k - transp;
    float v;
    if (p.flags & MPA_CTX_SEGM_TARGETS) {
      // dataset_context_segm's target is 4-D (1,1,seglength,n_out), so the reference's `y_trans[:, :, :transp] = 0`
      // (:273,276) clears |transp| *frames*, and the bins keep torch.roll's circular wrap -- reproduced, not fixed
      src %= p.n_out;
      if (src < 0) src += p.n_out;
      const bool cleared = p.n_out != 12 && ((transp > 0 && s < transp) || (transp < 0 && s >= p.seglength + transp));
      v = cleared ? 0.f : trow[src];
    } else if (p.n_out == 12) {
      src %= 12;
      if (src < 0) src += 12;
      v = trow[src];
    } else {
      v = (src >= 0 && src < p.n_out) ? trow[src] : 0.f;
    }
    p.y[i] = v;
  }
}


// 'aug:scalingfactor' (hcqt_datasets.py:211-225): the `seglength` frames between the context halves resampled to `new_len`
// frames by linear interpolation at linspace(0, seglength - 1, new_len) (scipy interp1d: slope and product in double, the
// difference of the two samples in float), the context halves copied.  out: (n_harm, hc + new_len + hc, n_bins).
__global__ __launch_bounds__(256) void time_scale_kernel(const float* __restrict__ src, long chan_stride, int n_harm, int n_bins,
                                                         int hc, int seglength, int new_len, float* __restrict__ out) {
  const int frames_out = new_len + 2 * hc;
  const long n = (long)n_harm * frames_out * n_bins;
  for (long e = blockIdx.x * 256L + threadIdx.x; e < n; e += (long)gridDim.x * 256) {
    const int b = (int)(e % n_bins);
    const int t = (int)((e / n_bins) % frames_out);
    const int c = (int)(e / ((long)n_bins * frames_out));
    const float* s = src + c * chan_stride + b;
    float v;
    if (t < hc) v = s[(long)t * n_bins];
    else if (t >= hc + new_len) v = s[(long)(t - new_len + seglength) * n_bins];
    else {
      const int k = t - hc;
      // numpy.linspace(0, L - 1, n): k * step with step = (L - 1) / (n - 1), the last point set to L - 1 exactly
      const double x = new_len > 1 ? (k == new_len - 1 ? (double)(seglength - 1) : k * ((double)(seglength - 1) / (double)(new_len - 1)))
                                   : 0.0;
      int lo = (int)x;                                  // interp1d: lo = searchsorted(x) - 1 clipped to [0, L - 2]
      if (x > 0.0 && (double)lo == x) lo -= 1;          // (side = 'left': an exact knot belongs to the interval below it)
      lo = max(0, min(lo, seglength - 2));
      if (seglength < 2) { v = s[(long)hc * n_bins]; }
      else {
        const float ylo = s[(long)(hc + lo) * n_bins], yhi = s[(long)(hc + lo + 1) * n_bins];
        const double slope = (double)(yhi - ylo) / 1.0;
        v = (float)(slope * (x - (double)lo) + (double)ylo);
      }
    }
    out[e] = v;
  }
}
}  // namespace

extern "C" int mpa_context_batch(const mpa_context_desc* d, int B, const uint64_t* src, const int64_t* chan_stride,
                                 const uint64_t* tgt, const int32_t* aug, const float* n1, const float* n2,
                                 const float* n3, uint64_t seed, float* X, float* y, void* stream) {
  if (!d || !src || !chan_stride || !tgt || !X || !y) return MPA_ERR_ARG;
  if (d->n_harm < 1 || d->n_harm > 16 || d->n_bins < 1 || d->n_bins > CTX_MAXBINS || d->frames < 1 || d->n_out < 1 || d->seglength < 1 || B < 0)
    return MPA_ERR_ARG;
  if ((d->flags & (MPA_CTX_EQ | MPA_CTX_TUNE | MPA_CTX_TRANSP)) && !aug) return MPA_ERR_ARG;
  if ((d->flags & MPA_CTX_TRANSP) && d->n_bins < 16) return MPA_ERR_ARG;
  if (B == 0) return MPA_OK;
  CtxParams p;
  p.n_harm = d->n_harm; p.n_bins = d->n_bins; p.frames = d->frames; p.n_out = d->n_out; p.seglength = d->seglength;
  p.flags = d->flags; p.B = B; p.compression = d->compression; p.noisestd = d->noisestd;
  for (int h = 0; h < 16; ++h) p.eq_off[h] = h == 0 ? -36 : (int)(36.0 * log2((double)h));   // hcqt_datasets.py:90-93
  p.src = src; p.chan_stride = chan_stride; p.tgt = tgt; p.aug = aug; p.n1 = n1; p.n2 = n2; p.n3 = n3; p.seed = seed;
  p.X = X; p.y = y;
  const long rows = (long)B * d->n_harm * d->frames;
  if (rows > 0x7fffffffL) return MPA_ERR_ARG;
  MPA_LAUNCH(context_rows_kernel, dim3((unsigned)((rows + CTX_ROWS - 1) / CTX_ROWS)), dim3(256), 0, (hipStream_t)stream,
             p, (int)rows);
  int rc = mpa_launch_status();
  if (rc != MPA_OK) return rc;
  const long n = (long)B * d->seglength * d->n_out;
  MPA_LAUNCH(context_targets_kernel, dim3((unsigned)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096)), dim3(256), 0,
             (hipStream_t)stream, p);
  return mpa_launch_status();
}

extern "C" int mpa_time_scale(const float* src, int64_t chan_stride, int n_harm, int n_bins, int half_context, int seglength,
                              int new_len, float* out, void* stream) {
  if (!src || !out || n_harm < 1 || n_bins < 1 || half_context < 0 || seglength < 1 || new_len < 1) return MPA_ERR_ARG;
  const long n = (long)n_harm * (new_len + 2L * half_context) * n_bins;
  MPA_LAUNCH(time_scale_kernel, dim3((unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192)), dim3(256), 0,
             (hipStream_t)stream, src, (long)chan_stride, n_harm, n_bins, half_context, seglength, new_len, out);
  return mpa_launch_status();
}
