// Evaluation measures of one recording on the GPU.
// Replaces libdl/metrics/eval_metrics.py:8-116 (calculate_single_measure, all 11 measures of exp180d...py:150-151)
// with libfmp/c5/c5s2_chord_rec_template.py:238-261 (compute_eval_measures) and
// libfmp/c3/c3s1_post_processing.py:60-68 (normalize_feature_sequence, norm '2').  Arithmetic in float64 like the
// reference's numpy code; inputs are the fp32 targets / network outputs.  HBM-bound and tiny; the two ranking measures
// sort the N*K scores once (rocPRIM device radix sort) and reduce over groups of tied scores.
#include "mpa_common.h"
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/functional.hpp>
#include <math.h>

namespace {

constexpr int NQ = 9;        // row-pass accumulators
constexpr int NR = 4;        // ranking accumulators: auc area, ap sum, positives, negatives
constexpr int MAXB = 1024;   // partial blocks

struct MaxOp {
  __device__ __forceinline__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; }
};

// one wave per frame; q: 0 TP, 1 #(pred>=thr), 2 #(targ>0), 3 sum cos, 4 sum bce terms, 5 sum euclid, 6 #(pred_thr==targ),
// 7 sum soft accuracy terms, 8 sum accumulated energy
__global__ __launch_bounds__(256) void eval_rows_kernel(const float* __restrict__ targ, const float* __restrict__ pred,
                                                        long nframes, int K, double thr, double* __restrict__ partial) {
  __shared__ double red[4][NQ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const double eps = 2.220446049250313e-16;            // np.finfo(float).eps (eval_metrics.py:49)
  double q[NQ];
#pragma unroll
  for (int i = 0; i < NQ; ++i) q[i] = 0.0;
  for (long n = (long)blockIdx.x * 4 + wave; n < nframes; n += (long)gridDim.x * 4) {
    const float* tr = targ + n * K;
    const float* pr = pred + n * K;
    double st = 0, sp = 0, stt = 0, spp = 0, stp = 0, sd = 0;
    for (int k = lane; k < K; k += 64) {
      const double t = (double)tr[k], p = (double)pr[k];
      const bool pt = p >= thr;
      q[0] += (t != 0.0 && pt) ? 1.0 : 0.0;
      q[1] += pt ? 1.0 : 0.0;
      q[2] += t > 0.0 ? 1.0 : 0.0;
      q[4] += t * log2(p + eps) + (1.0 - t) * log2(1.0 - p + eps);
      q[6] += ((pt ? 1.0 : 0.0) == t) ? 1.0 : 0.0;
      q[7] += t * p + (1.0 - t) * (1.0 - p);
      st += t; sp += p; stt += t * t; spp += p * p; stp += t * p; sd += (t - p) * (t - p);
    }
    st = mpa_wave_sum_d(st); sp = mpa_wave_sum_d(sp); stt = mpa_wave_sum_d(stt); spp = mpa_wave_sum_d(spp);
    stp = mpa_wave_sum_d(stp); sd = mpa_wave_sum_d(sd);
    if (lane == 0) {
      const double nt = sqrt(stt), np_ = sqrt(spp), rk = 1.0 / sqrt((double)K);
      const bool tok = nt > 1e-10, pok = np_ > 1e-10;                       // threshold_L2norm (eval_metrics.py:50)
      double c;
      if (tok && pok) c = stp / (nt * np_);
      else if (pok) c = sp / np_ * rk;                                     // silent target frame -> constant unit vector
      else if (tok) c = st / nt * rk;
      else c = 1.0;
      q[3] += c;
      q[5] += sqrt(sd);
      q[8] += stp / (st + eps);
    }
  }
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const double v = (i == 3 || i == 5 || i == 8) ? q[i] : mpa_wave_sum_d(q[i]);
    if (lane == 0) red[wave][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NQ)
    partial[(long)blockIdx.x * NQ + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// after the descending sort: 0/1 labels as counters and "first element of a group of tied scores" marks
__global__ __launch_bounds__(256) void rank_marks_kernel(const float* __restrict__ key, const float* __restrict__ lab,
                                                         uint32_t* __restrict__ lab_u, uint32_t* __restrict__ start, long n) {
  for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < n; j += (long)gridDim.x * 256) {
    lab_u[j] = lab[j] != 0.f ? 1u : 0u;
    start[j] = (j == 0 || key[j] != key[j - 1]) ? (uint32_t)j : 0u;
  }
}

// one term per group of tied scores (its last element i, its first element s = start[i]):
//   ROC trapezoid (fp_i - fp_s)(tp_i + tp_s)/2 and AP term (tp_i - tp_s) * tp_i/(tp_i+fp_i)
__global__ __launch_bounds__(256) void rank_groups_kernel(const float* __restrict__ key, const uint32_t* __restrict__ tps,
                                                          const uint32_t* __restrict__ start, long n,
                                                          double* __restrict__ partial) {
  __shared__ double red[4][2];
  double area = 0.0, ap = 0.0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    if (i != n - 1 && key[i] == key[i + 1]) continue;
    const long s = start[i];
    const double tp = (double)tps[i], fp = (double)(i + 1) - tp;
    const double tp0 = s > 0 ? (double)tps[s - 1] : 0.0, fp0 = (double)s - tp0;
    area += (fp - fp0) * (tp + tp0) * 0.5;
    ap += (tp - tp0) * (tp / (tp + fp));
  }
  area = mpa_wave_sum_d(area); ap = mpa_wave_sum_d(ap);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane == 0) { red[wave][0] = area; red[wave][1] = ap; }
  __syncthreads();
  if (threadIdx.x < 2)
    partial[(long)blockIdx.x * 2 + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// out[0..10]: the measures in the order of exp180d...py:150-151; out[11..13]: TP, FP, FN; out[14..15]: positives, negatives
__global__ __launch_bounds__(256) void eval_final_kernel(const double* __restrict__ prow, int nbrow,
                                                         const double* __restrict__ prank, int nbrank,
                                                         const uint32_t* __restrict__ tps, long nframes, int K,
                                                         double* __restrict__ out) {
  __shared__ double acc[NQ + 2];
  if (threadIdx.x < NQ + 2) {
    double s = 0.0;
    if (threadIdx.x < NQ) for (int b = 0; b < nbrow; ++b) s += prow[(long)b * NQ + threadIdx.x];
    else for (int b = 0; b < nbrank; ++b) s += prank[(long)b * 2 + (threadIdx.x - NQ)];
    acc[threadIdx.x] = s;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double n = (double)nframes, nk = n * (double)K;
    const double TP = acc[0], FP = acc[1] - TP, FN = acc[2] - TP;
    double P = 0.0, R = 0.0, F = 0.0;
    if (TP > 0.0) { P = TP / (TP + FP); R = TP / (TP + FN); F = 2.0 * P * R / (P + R); }   // c5s2...py:254-260
    const double pos = (double)tps[nframes * K - 1], neg = nk - pos;
    out[0] = P; out[1] = R; out[2] = F;
    out[3] = acc[3] / n;
    out[4] = -acc[4] / nk;
    out[5] = acc[5] / n;
    out[6] = acc[6] / nk;
    out[7] = acc[7] / nk;
    out[8] = acc[8] / n;
    out[9] = acc[NQ] / (pos * neg);        // NaN/inf when one class is absent: the host raises like scikit-learn
    out[10] = acc[NQ + 1] / pos;
    out[11] = TP; out[12] = FP; out[13] = FN; out[14] = pos; out[15] = neg;
  }
}

struct Layout {
  size_t key, lab, labu, tps, start, prow, prank, cub, cub_bytes, total;
};
static inline size_t up(size_t x) { return (x + 255) & ~(size_t)255; }

static int make_layout(long n, Layout& L) {
  size_t b_sort = 0, b_sum = 0, b_max = 0;
  const float* kf = nullptr; float* kfo = nullptr; const uint32_t* ui = nullptr; uint32_t* uo = nullptr;
  if (rocprim::radix_sort_pairs_desc(nullptr, b_sort, kf, kfo, kf, kfo, (size_t)n) != hipSuccess) return MPA_ERR_LAUNCH;
  if (rocprim::inclusive_scan(nullptr, b_sum, ui, uo, (size_t)n, rocprim::plus<uint32_t>()) != hipSuccess) return MPA_ERR_LAUNCH;
  if (rocprim::inclusive_scan(nullptr, b_max, ui, uo, (size_t)n, MaxOp()) != hipSuccess) return MPA_ERR_LAUNCH;
  size_t o = 0;
  L.key = o; o += up(n * 4);
  L.lab = o; o += up(n * 4);
  L.labu = o; o += up(n * 4);
  L.tps = o; o += up(n * 4);
  L.start = o; o += up(n * 4);
  L.prow = o; o += up((size_t)MAXB * NQ * 8);
  L.prank = o; o += up((size_t)MAXB * 2 * 8);
  L.cub = o;
  L.cub_bytes = b_sort > b_sum ? b_sort : b_sum;
  if (b_max > L.cub_bytes) L.cub_bytes = b_max;
  o += up(L.cub_bytes + 256);
  L.total = o;
  return MPA_OK;
}

}  // namespace

extern "C" int64_t mpa_eval_measures_workspace(int64_t n_frames, int n_bins) {
  if (n_frames < 1 || n_bins < 1 || n_frames * (int64_t)n_bins > 0x7fffffffLL) return MPA_ERR_ARG;
  Layout L;
  const int rc = make_layout((long)(n_frames * n_bins), L);
  return rc == MPA_OK ? (int64_t)L.total : rc;
}

extern "C" int mpa_eval_measures(const float* targ, const float* pred, int64_t n_frames, int n_bins, double threshold,
                                 double* out, void* ws, int64_t ws_bytes, void* stream) {
  if (!targ || !pred || !out || !ws) return MPA_ERR_ARG;
  if (n_frames < 1 || n_bins < 1 || n_frames * (int64_t)n_bins > 0x7fffffffLL) return MPA_ERR_ARG;
  const long n = (long)(n_frames * n_bins);
  Layout L;
  int rc = make_layout(n, L);
  if (rc != MPA_OK) return rc;
  if ((int64_t)L.total > ws_bytes) return MPA_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  char* w = (char*)ws;
  float* key = (float*)(w + L.key);
  float* lab = (float*)(w + L.lab);
  uint32_t* labu = (uint32_t*)(w + L.labu);
  uint32_t* tps = (uint32_t*)(w + L.tps);
  uint32_t* start = (uint32_t*)(w + L.start);
  double* prow = (double*)(w + L.prow);
  double* prank = (double*)(w + L.prank);
  void* cub = w + L.cub;
  size_t cb = L.cub_bytes;

  const int nbrow = (int)((n_frames + 3) / 4 < MAXB ? (n_frames + 3) / 4 : MAXB);
  MPA_LAUNCH(eval_rows_kernel, dim3(nbrow), dim3(256), 0, st, targ, pred, (long)n_frames, n_bins, threshold, prow);
  if ((rc = mpa_launch_status()) != MPA_OK) return rc;

  (void)hipGetLastError();
  if (rocprim::radix_sort_pairs_desc(cub, cb, pred, key, targ, lab, (size_t)n, 0, 32, st) != hipSuccess) return MPA_ERR_LAUNCH;
  const int nbel = (int)((n + 255) / 256 < MAXB ? (n + 255) / 256 : MAXB);
  MPA_LAUNCH(rank_marks_kernel, dim3(nbel), dim3(256), 0, st, key, lab, labu, start, n);
  if ((rc = mpa_launch_status()) != MPA_OK) return rc;
  cb = L.cub_bytes;
  if (rocprim::inclusive_scan(cub, cb, labu, tps, (size_t)n, rocprim::plus<uint32_t>(), st) != hipSuccess) return MPA_ERR_LAUNCH;
  cb = L.cub_bytes;
  if (rocprim::inclusive_scan(cub, cb, start, labu, (size_t)n, MaxOp(), st) != hipSuccess) return MPA_ERR_LAUNCH;
  MPA_LAUNCH(rank_groups_kernel, dim3(nbel), dim3(256), 0, st, key, tps, labu, n, prank);
  if ((rc = mpa_launch_status()) != MPA_OK) return rc;
  MPA_LAUNCH(eval_final_kernel, dim3(1), dim3(256), 0, st, prow, nbrow, prank, nbel, tps, (long)n_frames, n_bins, out);
  return mpa_launch_status();
}
