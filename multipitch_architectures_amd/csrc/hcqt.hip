// HCQT front-end on the GPU (second half): the pieces of compute_efficient_hcqt
// (libdl/data_preprocessing/hcqt.py:89-164) that are arithmetic.  ** PARITY UNPINNED **: the reference delegates to
// librosa 0.8 (librosa.cqt, librosa.estimate_tuning), which is absent from the image; these kernels implement the published
// algorithm as restated in oracle/restate_hcqt.py (direct evaluation of librosa's constant-Q filter bank at the original
// sample rate instead of its octave-by-octave resampling recursion) and are checked against that restatement only.
//   * the heavy part -- signal frames x filter bank -- runs as mpa_gemm with a strided A operand (A(m,k) = y[m*hop + k]:
//     frames overlap, nothing is copied) against a basis matrix generated on the device;
//   * STFT magnitude for the tuning estimate the same way (Hann-windowed DFT basis);
//   * piptrack / median / tuning histogram as small kernels (rocPRIM radix sort for the median).
#include "mpa_common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {

__global__ __launch_bounds__(256) void reflect_pad_kernel(const float* __restrict__ y, long n, long pad_l, long total,
                                                          float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long j = i - pad_l;                       // numpy's mode="reflect": period 2(n-1), no repeated edge sample
    if (n == 1) j = 0;
    else {
      const long period = 2 * (n - 1);
      j %= period;
      if (j < 0) j += period;
      if (j >= n) j = period - j;
    }
    out[i] = y[j];
  }
}

// B[k][2 f] = w[k] cos(2 pi k f / n_fft), B[k][2 f + 1] = -w[k] sin(...), w = periodic Hann
__global__ __launch_bounds__(256) void stft_basis_kernel(float* __restrict__ B, int n_fft) {
  const int nb = n_fft / 2 + 1;
  const long total = (long)n_fft * nb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i / nb), f = (int)(i - (long)k * nb);
    const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)k / (double)n_fft);
    const long r = ((long)k * f) % n_fft;                                   // exact phase reduction
    const double ph = 2.0 * M_PI * (double)r / (double)n_fft;
    B[(long)k * (2 * nb) + 2 * f] = (float)(w * cos(ph));
    B[(long)k * (2 * nb) + 2 * f + 1] = (float)(-w * sin(ph));
  }
}

__global__ __launch_bounds__(256) void complex_mag_kernel(const float* __restrict__ C, float* __restrict__ S, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float2 v = reinterpret_cast<const float2*>(C)[i];
    S[i] = sqrtf(v.x * v.x + v.y * v.y);
  }
}

// piptrack (librosa/core/pitch.py) on S [frames][nb]: one wave per frame; candidates (pitch, magnitude) appended to lists
__global__ __launch_bounds__(256) void piptrack_kernel(const float* __restrict__ S, long frames, int nb, double sr, int n_fft,
                                                       double fmin, double fmax, double threshold, double* __restrict__ pitch,
                                                       float* __restrict__ mag, int* __restrict__ count) {
  const long frame = (blockIdx.x * (long)blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (frame >= frames) return;
  const float* s = S + frame * nb;
  float mx = 0.f;
  for (int f = lane; f < nb; f += 64) mx = fmaxf(mx, s[f]);
  mx = mpa_wave_max(mx);
  const double ref = threshold * (double)mx;
  auto sm = [&](int f) { const double v = s[f]; return v > ref ? v : 0.0; };
  for (int f = lane; f < nb; f += 64) {
    const double fr = (double)f * sr / n_fft;
    if (!(fmin <= fr && fr < fmax) || f == 0) continue;
    bool loc;
    if (f == nb - 1) loc = sm(f) > sm(f - 1);
    else loc = sm(f) > sm(f - 1) && sm(f) >= sm(f + 1);
    if (!loc) continue;
    double avg = 0.0, shift = 0.0;
    if (f < nb - 1) {
      const double a = s[f - 1], b = s[f], c = s[f + 1];
      avg = 0.5 * (c - a);
      double den = 2.0 * b - c - a;
      if (fabs(den) < 2.2250738585072014e-308) den += 1.0;
      shift = avg / den;
    }
    const double p = ((double)f + shift) * sr / n_fft;
    const double m = (double)s[f] + 0.5 * avg * shift;
    if (p > 0.0) {
      const int k = atomicAdd(count, 1);
      pitch[k] = p;
      mag[k] = (float)m;
    }
  }
}

// tuning histogram over the candidates whose magnitude reaches the median (exact: the mean of the two middle values of an
// even count, as numpy); 100 bins of `resolution` over [-0.5, 0.5]
__global__ __launch_bounds__(256) void tuning_hist_kernel(const double* __restrict__ pitch, const float* __restrict__ mag,
                                                          const float* __restrict__ sorted, const int* __restrict__ count,
                                                          int bins_per_octave, int nbins, int* __restrict__ hist) {
  const int n = *count;
  if (n == 0) return;
  const double thr = (n & 1) ? (double)sorted[n / 2] : 0.5 * ((double)sorted[n / 2 - 1] + (double)sorted[n / 2]);
  const double step = 1.0 / nbins;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!((double)mag[i] >= thr)) continue;
    double r = bins_per_octave * log2(pitch[i] / 27.5);
    r = r - floor(r);
    if (r >= 0.5) r -= 1.0;
    int b = (int)floor((r + 0.5) * nbins);
    b = b < 0 ? 0 : (b > nbins - 1 ? nbins - 1 : b);
    while (b > 0 && r < -0.5 + b * step) --b;                       // np.histogram: half-open bins on linspace edges
    while (b < nbins - 1 && r >= -0.5 + (b + 1) * step) ++b;
    atomicAdd(hist + b, 1);
  }
}

__global__ void tuning_argmax_kernel(const int* __restrict__ hist, const int* __restrict__ count, int nbins,
                                     double* __restrict__ tuning) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (*count == 0) { *tuning = 0.0; return; }
  int best = 0;
  for (int b = 1; b < nbins; ++b)
    if (hist[b] > hist[best]) best = b;
  *tuning = -0.5 + best * (1.0 / nbins);
}

// constant-Q basis of a group of `nb` consecutive bins (librosa filters.constant_q: Hann-windowed complex exponentials of
// length Q sr / f, L1-normalised; response scaled by sqrt(length)): B[k][2 j], B[k][2 j + 1] = re, -im of bin j's filter at
// sample offset n = k - K0, zero outside its support; ncols/2 >= nb columns pairs (the rest zero)
__global__ __launch_bounds__(256) void cqt_basis_kernel(float* __restrict__ B, long K, long K0, int ncols, double f0, int nb,
                                                        int bins_per_octave, double sr) {
  const double Q = 1.0 / (exp2(1.0 / bins_per_octave) - 1.0);
  const long total = K * (ncols / 2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long k = i / (ncols / 2);
    const int j = (int)(i - k * (ncols / 2));
    float re = 0.f, im = 0.f;
    if (j < nb) {
      const double f = f0 * exp2((double)j / bins_per_octave);
      const double ilen = Q * sr / f;
      const double lo = floor(-ilen / 2.0), hi = floor(ilen / 2.0);      // np.arange(-ilen // 2, ilen // 2)
      const long M = (long)(hi - lo);
      const long n = k - K0;
      if ((double)n >= lo && (double)n < hi) {
        const long idx = n - (long)lo;
        const double w = 0.5 - 0.5 * cos(2.0 * M_PI * (double)idx / (double)M);
        // sum of the periodic Hann window over its M samples is exactly M / 2
        const double g = w / (0.5 * (double)M) * sqrt(ilen);
        double ph = f * (double)n / sr;
        ph -= floor(ph);
        re = (float)(g * cos(2.0 * M_PI * ph));
        im = (float)(-g * sin(2.0 * M_PI * ph));
      }
    }
    B[k * ncols + 2 * j] = re;
    B[k * ncols + 2 * j + 1] = im;
  }
}

// |C| of one bin group into the HCQT tensor out[n_bins_out][frames][n_harm]: bin `bin0 + j` of this CQT is row
// bin0 + j - fac_bins[m] of harmonic hidx[m] for every member m whose slice contains it (hcqt.py:158-162)
struct MagMembers { int n; int fac_bins[8]; int hidx[8]; };
__global__ __launch_bounds__(256) void cqt_mag_scatter_kernel(const float* __restrict__ C, long frames, int ncols, int nb,
                                                              int bin0, float* __restrict__ out, int n_bins_out, int n_harm,
                                                              MagMembers mm) {
  const long total = frames * nb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long t = i / nb;
    const int j = (int)(i - t * nb);
    const float2 v = *reinterpret_cast<const float2*>(C + t * ncols + 2 * j);
    const float m = sqrtf(v.x * v.x + v.y * v.y);
    for (int q = 0; q < mm.n; ++q) {
      const int row = bin0 + j - mm.fac_bins[q];
      if (row >= 0 && row < n_bins_out) out[((long)row * frames + t) * n_harm + mm.hidx[q]] = m;
    }
  }
}

inline unsigned grid_for(long n) { return (unsigned)std::min<long>(mpa_cdiv(n, 256), 4096); }

}  // namespace

extern "C" {

int mpa_reflect_pad(const float* y, int64_t n, int64_t pad_l, int64_t pad_r, float* out, void* stream) {
  if (!y || !out || n <= 0 || pad_l < 0 || pad_r < 0) return MPA_ERR_ARG;
  const long total = n + pad_l + pad_r;
  MPA_LAUNCH(reflect_pad_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, y, (long)n, (long)pad_l, total, out);
  return mpa_launch_status();
}

int mpa_stft_basis(float* B, int n_fft, void* stream) {
  if (!B || n_fft < 2 || (n_fft & 1)) return MPA_ERR_ARG;
  MPA_LAUNCH(stft_basis_kernel, dim3(grid_for((long)n_fft * (n_fft / 2 + 1))), dim3(256), 0, (hipStream_t)stream, B, n_fft);
  return mpa_launch_status();
}

int mpa_complex_mag(const float* C, float* S, int64_t n, void* stream) {
  if (!C || !S || n < 0) return MPA_ERR_ARG;
  if (n == 0) return MPA_OK;
  MPA_LAUNCH(complex_mag_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, C, S, (long)n);
  return mpa_launch_status();
}

int mpa_piptrack(const float* S, int64_t frames, int nb, double sr, int n_fft, double fmin, double fmax, double threshold,
                 double* pitch, float* mag, int* count, void* stream) {
  if (!S || !pitch || !mag || !count || frames <= 0 || nb != n_fft / 2 + 1) return MPA_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (mpa_zero_async(count, 4, s) != MPA_OK) return MPA_ERR_LAUNCH;
  MPA_LAUNCH(piptrack_kernel, dim3((unsigned)mpa_cdiv(frames, 4)), dim3(256), 0, s, S, (long)frames, nb, sr, n_fft, fmin, fmax,
             threshold, pitch, mag, count);
  return mpa_launch_status();
}

static size_t tuning_sort_temp(int64_t n) {
  size_t bytes = 0;
  (void)rocprim::radix_sort_keys(nullptr, bytes, (const float*)nullptr, (float*)nullptr, (size_t)std::max<int64_t>(n, 1));
  return bytes;
}

int64_t mpa_pitch_tuning_workspace(int64_t n) {
  if (n < 0) return MPA_ERR_ARG;
  return 256 + 4 * 256 + std::max<int64_t>(n, 1) * 4 + (int64_t)tuning_sort_temp(n) + 256;
}

int mpa_pitch_tuning(const double* pitch, const float* mag, int64_t n, int bins_per_octave, double resolution,
                     double* tuning_out, void* ws, int64_t ws_bytes, void* stream) {
  if (!tuning_out || !ws || n < 0 || resolution <= 0 || (n > 0 && (!pitch || !mag))) return MPA_ERR_ARG;
  if (ws_bytes < mpa_pitch_tuning_workspace(n)) return MPA_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int nbins = (int)ceil(1.0 / resolution);
  if (nbins > 256 || n > 0x7fffffffL) return MPA_ERR_UNSUPPORTED;
  char* p = (char*)ws;
  int* count = (int*)p; p += 256;
  int* hist = (int*)p; p += 4 * 256;
  float* sorted = (float*)p; p += std::max<int64_t>(n, 1) * 4;
  void* temp = p;
  if (mpa_zero_async(ws, 256 + 4 * 256, s) != MPA_OK) return MPA_ERR_LAUNCH;
  const int n32 = (int)n;
  if (hipMemcpyAsync(count, &n32, 4, hipMemcpyHostToDevice, s) != hipSuccess) return MPA_ERR_LAUNCH;
  if (n > 0) {
    size_t temp_bytes = tuning_sort_temp(n);
    if (rocprim::radix_sort_keys(temp, temp_bytes, mag, sorted, (size_t)n, 0, 32, s) != hipSuccess) return MPA_ERR_LAUNCH;
    MPA_LAUNCH(tuning_hist_kernel, dim3((unsigned)std::min<long>(mpa_cdiv(n, 256), 1024)), dim3(256), 0, s, pitch, mag,
               (const float*)sorted, (const int*)count, bins_per_octave, nbins, hist);
  }
  MPA_LAUNCH(tuning_argmax_kernel, dim3(1), dim3(64), 0, s, (const int*)hist, (const int*)count, nbins, tuning_out);
  return mpa_launch_status();
}

int mpa_cqt_basis(float* B, int64_t K, int64_t K0, int ncols, double f0, int nb, int bins_per_octave, double sr, void* stream) {
  if (!B || K <= 0 || K0 < 0 || ncols <= 0 || (ncols & 1) || nb <= 0 || 2 * nb > ncols || f0 <= 0 || sr <= 0) return MPA_ERR_ARG;
  MPA_LAUNCH(cqt_basis_kernel, dim3(grid_for(K * (ncols / 2))), dim3(256), 0, (hipStream_t)stream, B, (long)K, (long)K0, ncols, f0,
             nb, bins_per_octave, sr);
  return mpa_launch_status();
}

int mpa_cqt_mag_scatter(const float* C, int64_t frames, int ncols, int nb, int bin0, float* out, int n_bins_out, int n_harm,
                        const int* fac_bins, const int* hidx, int nmem, void* stream) {
  if (!C || !out || frames <= 0 || nb <= 0 || 2 * nb > ncols || nmem < 1 || nmem > 8 || !fac_bins || !hidx) return MPA_ERR_ARG;
  MagMembers mm{};
  mm.n = nmem;
  for (int q = 0; q < nmem; ++q) { mm.fac_bins[q] = fac_bins[q]; mm.hidx[q] = hidx[q]; }
  MPA_LAUNCH(cqt_mag_scatter_kernel, dim3(grid_for(frames * nb)), dim3(256), 0, (hipStream_t)stream, C, (long)frames, ncols, nb,
             bin0, out, n_bins_out, n_harm, mm);
  return mpa_launch_status();
}

}  // extern "C"
