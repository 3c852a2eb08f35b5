// Note list -> piano roll on the GPU (the pinnable half): replaces
// compute_annotation_array_nooverlap (libdl/data_preprocessing/hcqt.py:205-272), integer / index work, bit-exact.
//   annot_events_kernel  one workgroup: frame indices of every event (floor(t * fs) in float64, the reference's
//                        arithmetic), the corrections that keep every note >= 1 frame long (:239-257) and the row index
//                        (:263-268).  The corrections look sequential in the reference (a loop over the sorted unique end
//                        frames of the vanishing events that shifts every start / end equal to the current value) but each
//                        event only reads its own start / end: given the sorted list they are independent per event.
//   annot_paint_kernel   one wave per event writes 1.0 into [row][start:end) with numpy's slice semantics (:270).
#include "mpa_common.h"
#include <algorithm>

namespace {

constexpr int ANNOT_MAXV = 8192;      // vanishing events whose end frames fit the workgroup's LDS list

struct AnnotParams {
  const double* ev;      // [n][stride] start_sec, end_sec, pitch, ...
  int stride, n, kind, height, n_frames;
  double fs, shorten;
  int* sep;              // [n][3] start frame, end frame, row (workspace)
  int* status;           // 0 ok, 1 the reference's assertion ("still events of length<1"), 2 row out of bounds, 3 > ANNOT_MAXV
};

__global__ __launch_bounds__(1024) void annot_events_kernel(const AnnotParams p) {
  __shared__ long long vlist[ANNOT_MAXV];
  __shared__ long long vsorted[ANNOT_MAXV];
  __shared__ int nv, nu, bad;
  const int tid = threadIdx.x;
  if (tid == 0) { nv = 0; nu = 0; bad = 0; }
  __syncthreads();
  // pass 1: end frames of the vanishing events
  for (int i = tid; i < p.n; i += blockDim.x) {
    const double t0 = p.ev[(long)i * p.stride], t1r = p.ev[(long)i * p.stride + 1];
    const double t1 = p.shorten != 1.0 ? t0 + p.shorten * (t1r - t0) : t1r;
    const long long s = (long long)floor(t0 * p.fs), e = (long long)floor(t1 * p.fs);
    if (e - s < 1) {
      const int k = atomicAdd(&nv, 1);
      if (k < ANNOT_MAXV) vlist[k] = e;
    }
  }
  __syncthreads();
  if (nv > ANNOT_MAXV) { if (tid == 0) p.status[0] = 3; return; }
  // sorted unique: rank sort (ties by position), then one thread drops the duplicates
  const int m = nv;
  for (int j = tid; j < m; j += blockDim.x) {
    const long long v = vlist[j];
    int r = 0;
    for (int k = 0; k < m; ++k) r += (vlist[k] < v) || (vlist[k] == v && k < j);
    vsorted[r] = v;
  }
  __syncthreads();
  if (tid == 0) {
    int u = 0;
    for (int k = 0; k < m; ++k)
      if (k == 0 || vsorted[k] != vsorted[k - 1]) vlist[u++] = vsorted[k];
    nu = u;
  }
  __syncthreads();
  const int mu = nu;
  // pass 2: every event through the corrections
  for (int i = tid; i < p.n; i += blockDim.x) {
    const double t0 = p.ev[(long)i * p.stride], t1r = p.ev[(long)i * p.stride + 1], pit = p.ev[(long)i * p.stride + 2];
    const double t1 = p.shorten != 1.0 ? t0 + p.shorten * (t1r - t0) : t1r;
    long long s = (long long)floor(t0 * p.fs), e = (long long)floor(t1 * p.fs);
    const bool gone = e - s < 1;
    for (int k = 0; k < mu; ++k) {          // ascending: a later value sees the shifts of the earlier ones
      const long long v = vlist[k];
      if (s == v) s += 1;
      if (e == v) e += 1;
    }
    if (gone) s -= 1;
    if (e - s < 1) s -= 1;
    if (e - s < 1) atomicOr(&bad, 1);
    long long row;
    if (p.kind == 0) {                      // pitch class: int(np.mod(pitch, 12)) -- floored modulo
      const double r = pit - 12.0 * floor(pit / 12.0);
      row = (long long)r;
    } else if (p.kind == 1) row = (long long)pit;     // int(): truncation
    else row = 0;
    if (row >= p.height || row < -(long long)p.height) atomicOr(&bad, 2);
    if (row < 0) row += p.height;
    // numpy slice semantics for [s:e) on an axis of n_frames
    const long long L = p.n_frames;
    long long a = s < 0 ? s + L : s, b = e < 0 ? e + L : e;
    a = a < 0 ? 0 : (a > L ? L : a);
    b = b < 0 ? 0 : (b > L ? L : b);
    p.sep[3 * (long)i] = (int)a;
    p.sep[3 * (long)i + 1] = (int)b;
    p.sep[3 * (long)i + 2] = (int)row;
  }
  __syncthreads();
  if (tid == 0) p.status[0] = (bad & 1) ? 1 : ((bad & 2) ? 2 : 0);
}

__global__ __launch_bounds__(256) void annot_paint_kernel(const int* __restrict__ sep, const int* __restrict__ status, int n,
                                                          int n_frames, double* __restrict__ out) {
  if (status[0] != 0) return;             // the caller raises; nothing is painted
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n; i += nwaves) {
    const int a = sep[3 * i], b = sep[3 * i + 1], row = sep[3 * i + 2];
    double* dst = out + (long)row * n_frames;
    for (int j = a + lane; j < b; j += 64) dst[j] = 1.0;
  }
}

}  // namespace

extern "C" {

int64_t mpa_annotation_workspace(int n_events) { return n_events < 0 ? MPA_ERR_ARG : (int64_t)(3 * (int64_t)n_events + 4) * 4; }

int mpa_annotation_array_nooverlap(const double* note_events, int ev_stride, int n_events, double fs_hcqt, double shorten,
                                   int kind, int n_frames, double* out, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
  if (!note_events || !out || !workspace || n_events < 0 || ev_stride < 3 || n_frames < 0 || kind < 0 || kind > 2)
    return MPA_ERR_ARG;
  if (workspace_bytes < mpa_annotation_workspace(n_events)) return MPA_ERR_WORKSPACE;
  const int height = kind == 0 ? 12 : (kind == 1 ? 128 : 1);
  hipStream_t s = (hipStream_t)stream;
  AnnotParams p{};
  p.ev = note_events; p.stride = ev_stride; p.n = n_events; p.kind = kind; p.height = height; p.n_frames = n_frames;
  p.fs = fs_hcqt; p.shorten = shorten;
  p.status = (int*)workspace;
  p.sep = (int*)workspace + 4;
  if (mpa_zero_async(out, sizeof(double) * (size_t)height * (size_t)n_frames, s) != MPA_OK) return MPA_ERR_LAUNCH;
  MPA_LAUNCH(annot_events_kernel, dim3(1), dim3(1024), 0, s, p);
  int rc = mpa_launch_status();
  if (rc) return rc;
  if (n_events == 0) return MPA_OK;
  const int blocks = (int)std::min<long>(mpa_cdiv(n_events, 4), 1024);
  MPA_LAUNCH(annot_paint_kernel, dim3(blocks), dim3(256), 0, s, (const int*)p.sep, (const int*)p.status, n_events, n_frames, out);
  return mpa_launch_status();
}

}  // extern "C"
