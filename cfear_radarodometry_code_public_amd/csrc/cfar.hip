// cfar.hip -- azimuth CA-CFAR, the alternative stage-1 filter of radarDriver::Process
// (radar_driver.cpp:52-56: AzimuthCACFAR(window_size, false_alarm_rate, nb_guard_cells, range_res, z_min,
// min_distance, 400.0).getFilteredPointCloud, cfar.cpp:27-87).
//
// Per range bin that passes the static test, the detector compares the squared intensity with a scaled mean of the
// squared intensities in a trailing and a forwarding window (guard cells in between). One 256-thread workgroup
// owns one azimuth row: the row is staged in LDS with aligned dword loads, the windows become differences of an
// LDS prefix sum of squares (integers, exact), and the decision replays the reference's double arithmetic
// (sum / N per window, (t + f) / 2, scaling * mean, I^2 > threshold; an empty window gives 0/0 = NaN and no
// detection). The output cloud is row-major over (azimuth, range bin) like the reference's push_back order:
// pass 1 counts per row, a scan turns the counts into row offsets, pass 2 writes.
#include <math.h>

#include "blockops.h"
#include "common.h"

namespace {
using namespace cfear_dev;

constexpr int CFAR_BLOCK = 256;
constexpr int CFAR_MAX_R = 16384;  // range bins per azimuth the LDS row / prefix arrays hold

struct CfarParams {
  int A, R, window, guard;
  double range_res, static_threshold, min_distance, max_distance, scaling;
};

template <bool EMIT>
__global__ __launch_bounds__(CFAR_BLOCK) void cfar_kernel(const uint8_t* __restrict__ polar, long long alloc_bytes, CfarParams P,
                                                          const double* __restrict__ trig, int* __restrict__ row_count,
                                                          const int* __restrict__ row_base, float* __restrict__ xyi, int cap) {
  __shared__ uint32_t prefix[CFAR_MAX_R + 1];                        // prefix[i] = sum of squares of bins < i
  __shared__ __attribute__((aligned(16))) uint8_t rowbuf[CFAR_MAX_R + 8];
  __shared__ int red_i[64];
  // batches: image blockIdx.x / A, azimuth blockIdx.x % A; the images lie back to back, so row blockIdx.x starts at blockIdx.x * R
  const int grow = blockIdx.x, img = grow / P.A, az = grow - img * P.A, tid = threadIdx.x, R = P.R;
  // ---- stage the row with aligned dword loads (the bytes around the row belong to the neighbouring rows) ----
  const long long row_off = (long long)grow * R;
  const int first = (int)(row_off & 3);
  const long long base = row_off - first;
  const int ndw = (first + R + 3) >> 2;
  for (int i = tid; i < ndw; i += CFAR_BLOCK) {
    const long long o = base + 4LL * i;
    uint32_t v;
    if (o + 4 <= alloc_bytes) {
      v = *reinterpret_cast<const uint32_t*>(polar + o);
    } else {  // last dword of the allocation: bytewise
      v = 0;
      for (int b = 0; b < 4; b++)
        if (o + b < alloc_bytes) v |= (uint32_t)polar[o + b] << (8 * b);
    }
    reinterpret_cast<uint32_t*>(rowbuf)[i] = v;
  }
  __syncthreads();
  const uint8_t* row = rowbuf + first;
  // ---- prefix sum of squares: consecutive bins per thread ----
  const int ipt = (R + CFAR_BLOCK - 1) / CFAR_BLOCK;
  const int b0 = tid * ipt, b1 = min(R, b0 + ipt);
  {
    int s = 0;
    for (int i = b0; i < b1; i++) { const int v = row[i]; s += v * v; }
    int tot;
    int o = block_exclusive_scan(s, red_i, &tot);
    for (int i = b0; i < b1; i++) { prefix[i] = (uint32_t)o; const int v = row[i]; o += v * v; }
    if (tid == 0) prefix[R] = (uint32_t)tot;
    __syncthreads();
  }
  // ---- detections of this thread's bins (ascending) ----
  const double cos_t = trig[2 * az], sin_t = trig[2 * az + 1];  // theta = (az + 1) / A * 2 pi, host libm (cfar.cpp:40)
  int cnt = 0;
  unsigned long long hit = 0;  // ipt <= 64 bins per thread (R <= 16384)
  for (int i = b0; i < b1; i++) {
    const double range = P.range_res * (double)i;
    const int iv = row[i];
    const double intensity = (double)iv;
    if (range > P.min_distance && range < P.max_distance && intensity > P.static_threshold) {  // cfar.cpp:45
      const int t0 = max(0, i - P.guard - P.window), t1 = i - P.guard;                           // :48-49
      const int f0 = i + P.guard, f1 = min(R, i + P.guard + P.window);                           // :52-53
      const double tn = t1 > t0 ? (double)(t1 - t0) : 0.0, fn = f1 > f0 ? (double)(f1 - f0) : 0.0;
      const double ts = t1 > t0 ? (double)(prefix[t1] - prefix[t0]) : 0.0;
      const double fs = f1 > f0 ? (double)(prefix[f1] - prefix[f0]) : 0.0;
      const double mean = (ts / tn + fs / fn) / 2.0;  // empty window: 0/0 = NaN -> no detection (:56)
      const double threshold = P.scaling * mean;
      if ((double)(iv * iv) > threshold) { hit |= 1ull << (i - b0); cnt++; }                     // :58-60
    }
  }
  int total;
  int o = block_exclusive_scan(cnt, red_i, &total);
  if (!EMIT) {
    if (tid == 0) row_count[grow] = total;
    return;
  }
  o += row_base[grow];
  xyi += 3 * (size_t)img * cap;  // image i writes at most `cap` points at xyi + i * cap * 3
  while (hit) {
    const int b = __ffsll((long long)hit) - 1;
    hit &= hit - 1;
    const int i = b0 + b;
    if (o < cap) {
      const double range = P.range_res * (double)i;
      xyi[3 * (size_t)o + 0] = (float)(range * cos_t);  // :63-65
      xyi[3 * (size_t)o + 1] = (float)(range * sin_t);
      xyi[3 * (size_t)o + 2] = (float)row[i];
    }
    o++;
  }
}

// exclusive scan of the A row counts of image blockIdx.x (one workgroup per image); total -> d_total[blockIdx.x]
__global__ __launch_bounds__(1024) void cfar_row_scan_kernel(const int* __restrict__ row_count, int A, int* __restrict__ row_base,
                                                             int* __restrict__ d_total) {
  __shared__ int red_i[64];
  row_count += (size_t)blockIdx.x * A; row_base += (size_t)blockIdx.x * A; d_total += blockIdx.x;
  const int ipt = (A + blockDim.x - 1) / blockDim.x;
  const int i0 = threadIdx.x * ipt, i1 = min(A, i0 + ipt);
  int s = 0;
  for (int i = i0; i < i1; i++) s += row_count[i];
  int tot;
  int o = block_exclusive_scan(s, red_i, &tot);
  for (int i = i0; i < i1; i++) { row_base[i] = o; o += row_count[i]; }
  if (threadIdx.x == 0) *d_total = tot;
}

int cfar_run(cfear_ctx* ctx, const uint8_t* d_polar, long long alloc_bytes, int window_size, int nb_guard_cells,
             float false_alarm_rate, double max_distance, cfear_cloud** out) {
  if (!out) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: null output");
  *out = nullptr;
  if (window_size < 1 || nb_guard_cells < 0 || !(false_alarm_rate > 0.f))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: window_size >= 1, nb_guard_cells >= 0, false_alarm_rate > 0 required");
  if (ctx->R > CFAR_MAX_R) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar: more than 16384 range bins");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  CfarParams P;
  P.A = ctx->A; P.R = ctx->R; P.window = window_size; P.guard = nb_guard_cells;
  // float members of radarDriver::Parameters bound to const double& (radar_driver.cpp:54)
  P.range_res = (double)ctx->par.range_res; P.static_threshold = (double)ctx->par.z_min; P.min_distance = (double)ctx->par.min_distance;
  P.max_distance = max_distance;
  const double N = (double)(window_size * 2);  // CFARFilter::getCAScalingFactor (cfar.cpp:12-16, :32), host libm
  P.scaling = N * (pow((double)false_alarm_rate, -1. / N) - 1.);
  int* d_tmp = nullptr;  // row counts, row bases, total
  if (hipMalloc(&d_tmp, sizeof(int) * (2 * (size_t)P.A + 1)) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc cfar rows");
  int* d_count = d_tmp; int* d_base = d_tmp + P.A; int* d_total = d_tmp + 2 * P.A;
  hipLaunchKernelGGL((cfar_kernel<false>), dim3(P.A), dim3(CFAR_BLOCK), 0, ctx->stream, d_polar, alloc_bytes, P, ctx->d_trig, d_count,
                     d_base, (float*)nullptr, 0);
  hipLaunchKernelGGL(cfar_row_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_count, P.A, d_base, d_total);
  int total = 0;
  hipError_t e = hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { (void)hipFree(d_tmp); return cfear_fail(ctx, CFEAR_ERR_HIP, "filter_cfar count pass", e); }
  cfear_cloud* c = nullptr;
  int rc = cfear_cloud_alloc(ctx, total, &c);
  if (rc != CFEAR_OK) { (void)hipFree(d_tmp); return rc; }
  if (total > 0)
    hipLaunchKernelGGL((cfar_kernel<true>), dim3(P.A), dim3(CFAR_BLOCK), 0, ctx->stream, d_polar, alloc_bytes, P, ctx->d_trig, d_count,
                       d_base, c->d_xyi, c->cap);
  e = hipMemcpyAsync(c->d_n, d_total, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_tmp);
  if (e != hipSuccess) { cfear_cloud_release(ctx, c); return cfear_fail(ctx, CFEAR_ERR_HIP, "filter_cfar emit pass", e); }
  *out = c;
  return CFEAR_OK;
}

}  // namespace

// the batched filter on `stream` with the caller's row scratch (2 * n_scans * A ints: row counts, row bases): count pass, one row
// scan per image, emit pass (cfear_filter_cfar_batch_device; the CA-CFAR stage of the batched odometry objects, pipeline.hip)
__attribute__((visibility("hidden"))) int cfear_launch_cfar_batch(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, int window_size, int nb_guard_cells,
                                                                  float false_alarm_rate, double max_distance, float* d_xyi, int capacity, int* d_counts,
                                                                  int* d_rows, hipStream_t stream) {
  if ((reinterpret_cast<uintptr_t>(d_polar) & 3) != 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar_batch: polar buffer must be 4-byte aligned");
  if (window_size < 1 || nb_guard_cells < 0 || !(false_alarm_rate > 0.f))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar_batch: window_size >= 1, nb_guard_cells >= 0, false_alarm_rate > 0 required");
  if (ctx->R > CFAR_MAX_R) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar_batch: more than 16384 range bins");
  if ((long long)n_scans * ctx->A > 0x7FFFFFFFLL) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar_batch: too many rows");
  const size_t rows = (size_t)n_scans * ctx->A;
  CfarParams P;
  P.A = ctx->A; P.R = ctx->R; P.window = window_size; P.guard = nb_guard_cells;
  P.range_res = (double)ctx->par.range_res; P.static_threshold = (double)ctx->par.z_min; P.min_distance = (double)ctx->par.min_distance;
  P.max_distance = max_distance;
  const double N = (double)(window_size * 2);
  P.scaling = N * (pow((double)false_alarm_rate, -1. / N) - 1.);
  int* d_count = d_rows; int* d_base = d_count + rows;
  const long long alloc = (long long)rows * ctx->R;
  hipLaunchKernelGGL((cfar_kernel<false>), dim3((unsigned)rows), dim3(CFAR_BLOCK), 0, stream, d_polar, alloc, P, ctx->d_trig, d_count, d_base,
                     (float*)nullptr, 0);
  hipLaunchKernelGGL(cfar_row_scan_kernel, dim3(n_scans), dim3(1024), 0, stream, d_count, P.A, d_base, d_counts);
  hipLaunchKernelGGL((cfar_kernel<true>), dim3((unsigned)rows), dim3(CFAR_BLOCK), 0, stream, d_polar, alloc, P, ctx->d_trig, d_count, d_base,
                     d_xyi, capacity);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

extern "C" {

int cfear_filter_cfar_device(cfear_ctx* ctx, const uint8_t* d_polar, int window_size, int nb_guard_cells, float false_alarm_rate,
                             double max_distance, cfear_cloud** cloud) {
  if (!ctx || !d_polar) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: bad argument");
  if ((reinterpret_cast<uintptr_t>(d_polar) & 3) != 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: polar buffer must be 4-byte aligned");
  return cfar_run(ctx, d_polar, (long long)ctx->A * ctx->R, window_size, nb_guard_cells, false_alarm_rate, max_distance, cloud);
}

int cfear_filter_cfar_batch_device(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, int window_size, int nb_guard_cells,
                                   float false_alarm_rate, double max_distance, float* d_xyi, int capacity, int* d_counts) {
  if (!ctx || !d_polar || !d_xyi || !d_counts || n_scans <= 0 || capacity <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar_batch: bad argument");
  if ((long long)n_scans * ctx->A > 0x7FFFFFFFLL) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar_batch: too many rows");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t rows = (size_t)n_scans * ctx->A;
  if (2 * rows > ctx->cfar_rows_cap) {  // row counts and row bases of the whole batch
    if (ctx->d_cfar_rows) { CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_cfar_rows); }
    ctx->d_cfar_rows = nullptr; ctx->cfar_rows_cap = 0;
    if (hipMalloc(&ctx->d_cfar_rows, sizeof(int) * 2 * rows) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc cfar rows");
    ctx->cfar_rows_cap = 2 * rows;
  }
  return cfear_launch_cfar_batch(ctx, d_polar, n_scans, window_size, nb_guard_cells, false_alarm_rate, max_distance, d_xyi, capacity, d_counts,
                                 ctx->d_cfar_rows, ctx->stream);
}

int cfear_filter_cfar(cfear_ctx* ctx, const uint8_t* h_polar, int window_size, int nb_guard_cells, float false_alarm_rate,
                      double max_distance, cfear_cloud** cloud) {
  if (!ctx || !h_polar) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = cfear_ensure_staging(ctx, 1);
  if (rc != CFEAR_OK) return rc;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_polar, h_polar, (size_t)ctx->A * ctx->R, hipMemcpyHostToDevice, ctx->stream));
  return cfar_run(ctx, ctx->d_polar, (long long)ctx->A * ctx->R, window_size, nb_guard_cells, false_alarm_rate, max_distance, cloud);
}

}  // extern "C"
