// cabi.hip -- C ABI of libcfear_hip.so: context management + stage-1 entry points.
// Interface contract and reference citations: include/cfear_hip.h.
#include <math.h>

#include "common.h"

extern "C" {

const char* cfear_version(void) { return "cfear-hip 0.1 (gfx950)"; }

void cfear_default_params(cfear_params* p) {
  memset(p, 0, sizeof(*p));
  p->z_min = 60.f;            // radar_driver.h:40
  p->range_res = 0.0438f;     // radar_driver.h:41
  p->min_distance = 2.5f;     // radar_driver.h:45
  p->k_strongest = 12;        // radar_driver.h:42
  p->res = 3.0;               // odometrykeyframefuser.h:132
  p->downsample_factor = 1.0; // pointnormal.cpp:5
  p->weight_intensity = 1;    // offline_odometry.cpp:160
  p->cost = CFEAR_COST_P2L;   // odometrykeyframefuser.h:86
  p->loss = CFEAR_LOSS_HUBER; // odometrykeyframefuser.h:99
  p->weight_opt = 4;          // launch/oxford/eval/params/baseline/*
  p->loss_limit = 0.1;
  p->covar_scale = 1.0;
  p->regularization = 0.1;
  p->submap_scan_size = 4;
  p->compensate = 1;
  p->radar_ccw = 0;
  p->use_keyframe = 1;
  p->min_keyframe_dist = 1.5;
  p->min_keyframe_rot_deg = 5.0;
  p->max_itr_association = 8;   // n_scan_normal.h:75
  p->min_itr = 3;               // n_scan_normal.h:75
  p->max_solver_iterations = 20; // n_scan_normal.cpp:9
  p->assoc_radius = 2.0;        // registration.h:122
  p->filter_type = CFEAR_FILTER_KSTRONG;  // radar_driver.h:48
  p->cfar_window_size = 10; p->cfar_nb_guard_cells = 20;  // radar_driver.h:43
  p->cfar_false_alarm_rate = 0.01f;                       // radar_driver.h:44
  p->cfar_max_distance = 400.0;                           // radar_driver.cpp:54
  p->cfar_max_points = 0;
}

static int validate_params(cfear_ctx* ctx, const cfear_params* p) {
  if (p->k_strongest < 1 || p->k_strongest > 64) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "k_strongest must be in 1..64");
  if (!(p->range_res > 0.f)) return cfear_fail(ctx, CFEAR_ERR_INVALID, "range_res must be > 0");
  if (!(p->res > 0.05)) return cfear_fail(ctx, CFEAR_ERR_INVALID, "res must be > 0.05 (odometrykeyframefuser.cpp:25)");
  if (!(p->downsample_factor > 0)) return cfear_fail(ctx, CFEAR_ERR_INVALID, "downsample_factor must be > 0");
  if (p->submap_scan_size < 1 || p->submap_scan_size > 63) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "submap_scan_size must be in 1..63");
  if (p->max_itr_association < 1 || p->max_itr_association > CFEAR_MAX_OUTER) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "max_itr_association must be in 1..64");
  if (p->cost < 0 || p->cost > 2) return cfear_fail(ctx, CFEAR_ERR_INVALID, "unknown cost");
  if (p->loss < 0 || p->loss > 5) return cfear_fail(ctx, CFEAR_ERR_INVALID, "unknown loss");
  // the cell-mean grid is sized by assoc_radius (a non-positive or non-finite value would never let the grid fit)
  if (!(p->assoc_radius > 0) || !isfinite(p->assoc_radius)) return cfear_fail(ctx, CFEAR_ERR_INVALID, "assoc_radius must be finite and > 0");
  if (p->max_solver_iterations < 1) return cfear_fail(ctx, CFEAR_ERR_INVALID, "max_solver_iterations must be >= 1");
  if (p->filter_type != CFEAR_FILTER_KSTRONG && p->filter_type != CFEAR_FILTER_CACFAR) return cfear_fail(ctx, CFEAR_ERR_INVALID, "unknown filter_type");
  if (p->filter_type == CFEAR_FILTER_CACFAR && (p->cfar_window_size < 1 || p->cfar_nb_guard_cells < 0 || !(p->cfar_false_alarm_rate > 0.f) || p->cfar_max_points < 0))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "CA-CFAR: cfar_window_size >= 1, cfar_nb_guard_cells >= 0, cfar_false_alarm_rate > 0, cfar_max_points >= 0 required");
  if (p->min_itr < 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "min_itr must be >= 0");
  if (!(p->min_distance >= 0.f) || !isfinite(p->min_distance)) return cfear_fail(ctx, CFEAR_ERR_INVALID, "min_distance must be finite and >= 0");
  if (!isfinite(p->z_min) || !isfinite(p->range_res) || !isfinite(p->res) || !isfinite(p->loss_limit))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "z_min, range_res, res and loss_limit must be finite");
  return CFEAR_OK;
}

int cfear_create(cfear_ctx** out, int device, void* stream, const cfear_params* p, int A, int R) {
  if (!out || !p || A <= 0 || R <= 0) return CFEAR_ERR_INVALID;
  *out = nullptr;
  cfear_ctx* ctx = new cfear_ctx();
  ctx->device = device;
  int rc = validate_params(ctx, p);
  if (rc != CFEAR_OK) { delete ctx; return rc; }
  if (R + 27 > 16 * 1024) { delete ctx; return CFEAR_ERR_UNSUPPORTED; }
  ctx->par = *p;
  ctx->A = A;
  ctx->R = R;
  hipError_t e = hipSetDevice(device);
  if (e != hipSuccess) { delete ctx; return CFEAR_ERR_HIP; }
  if (stream) {
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return CFEAR_ERR_HIP; }
    ctx->own_stream = true;
  }
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) { cfear_destroy(ctx); return CFEAR_ERR_HIP; }
  // theta = (double(bearing + 1) / nb_azimuths) * 2 * M_PI (radar_filters.cpp:317), cos/sin by host libm
  std::vector<double> trig(2 * (size_t)A);
  for (int b = 0; b < A; b++) {
    const double theta = ((double)(b + 1) / A) * 2. * M_PI;
    trig[2 * b] = cos(theta);
    trig[2 * b + 1] = sin(theta);
  }
  if (hipMalloc(&ctx->d_trig, trig.size() * sizeof(double)) != hipSuccess) { cfear_destroy(ctx); return CFEAR_ERR_NOMEM; }
  if (hipMemcpy(ctx->d_trig, trig.data(), trig.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { cfear_destroy(ctx); return CFEAR_ERR_HIP; }
  *out = ctx;
  return CFEAR_OK;
}

void cfear_destroy(cfear_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->d_trig) (void)hipFree(ctx->d_trig);
  if (ctx->d_polar) (void)hipFree(ctx->d_polar);
  if (ctx->d_slots) (void)hipFree(ctx->d_slots);
  if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
  if (ctx->d_cfar_rows) (void)hipFree(ctx->d_cfar_rows);
  for (auto& b : ctx->pool) (void)hipFree(b.second);
  for (auto& b : ctx->hpool) (void)hipHostFree(b.second);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->h_img) (void)hipHostFree(ctx->h_img);
  if (ctx->ev_img) (void)hipEventDestroy(ctx->ev_img);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* cfear_last_error(const cfear_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int cfear_set_params(cfear_ctx* ctx, const cfear_params* p) {
  if (!ctx || !p) return CFEAR_ERR_INVALID;
  int rc = validate_params(ctx, p);
  if (rc != CFEAR_OK) return rc;
  ctx->par = *p;
  return CFEAR_OK;
}

int cfear_tune(cfear_ctx* ctx, int key, int value) {
  if (!ctx) return CFEAR_ERR_INVALID;
  switch (key) {
    case CFEAR_TUNE_FILTER_OCCUPANCY: ctx->tune_k1_occ = value; return CFEAR_OK;
    case CFEAR_TUNE_FILTER_ROWS_PER_WAVE: ctx->tune_k1_rows = value > 0 ? value : 0; return CFEAR_OK;
    case CFEAR_TUNE_ODOMETRY_OVERLAP: ctx->tune_odo_overlap = value < 0 ? 0 : (value > 8 ? 8 : value); return CFEAR_OK;
    case CFEAR_TUNE_FILTER_CUS: ctx->tune_filter_cus = value < 0 ? 0 : value; return CFEAR_OK;
    case CFEAR_TUNE_REGISTRATION_ORDER: ctx->tune_reg_order = value != 0; return CFEAR_OK;
    case CFEAR_TUNE_MAX_CELLS: ctx->tune_max_cells = value < 0 ? 0 : value; return CFEAR_OK;
    case CFEAR_TUNE_REPEAT_SHORTCUT: ctx->tune_repeat_shortcut = value != 0; return CFEAR_OK;
    case CFEAR_TUNE_VOXEL_ORDER: ctx->tune_voxel_order = value == 1 ? 1 : 0; return CFEAR_OK;
    case CFEAR_TUNE_NN_TIE_RULE: ctx->tune_nn_tie = (value < 0 || value > 2) ? 0 : value; return CFEAR_OK;
    case CFEAR_TUNE_LARGE_SUBMAP_KERNEL: ctx->tune_large_kernel = (value < 0 || value > 2) ? 0 : value; return CFEAR_OK;
    case CFEAR_TUNE_REPLAY_PERSISTENT_MAX: ctx->tune_replay_persistent_max = value < 0 ? 0 : value; return CFEAR_OK;
    default: return cfear_fail(ctx, CFEAR_ERR_INVALID, "tune: unknown key");
  }
}

int cfear_synchronize(cfear_ctx* ctx) {
  if (!ctx) return CFEAR_ERR_INVALID;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (hipStream_t st : ctx->aux_streams) CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(st));
  return CFEAR_OK;
}

int cfear_kstrongest_device(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, uint32_t* d_slots) {
  if (!ctx) return CFEAR_ERR_INVALID;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return cfear_launch_kstrongest(ctx, d_polar, n_scans, d_slots, ctx->stream);
}

// ---- block pool of the per-call handles (clouds, scans) and the pinned staging of their downloads (common.h) ----
int cfear_pool_alloc(cfear_ctx* ctx, size_t bytes, void** out, size_t* got) {
  const size_t need = (bytes + 4095) & ~(size_t)4095;
  int best = -1;
  for (int i = 0; i < (int)ctx->pool.size(); i++)  // smallest parked block that fits without wasting more than half of it
    if (ctx->pool[i].first >= need && ctx->pool[i].first <= 2 * need + 65536 && (best < 0 || ctx->pool[i].first < ctx->pool[best].first)) best = i;
  if (best >= 0) {
    *out = ctx->pool[best].second; *got = ctx->pool[best].first;
    ctx->pool_bytes -= ctx->pool[best].first;
    ctx->pool[best] = ctx->pool.back(); ctx->pool.pop_back();
    return CFEAR_OK;
  }
  if (hipMalloc(out, need) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc (handle block)");
  *got = need;
  return CFEAR_OK;
}
void cfear_pool_free(cfear_ctx* ctx, void* p, size_t bytes) {
  if (!p) return;
  constexpr size_t POOL_MAX_BYTES = (size_t)512 << 20;
  constexpr size_t POOL_MAX_BLOCKS = 256;
  if (!ctx) { (void)hipFree(p); return; }
  // work on other streams of this context may still read the block: the old behaviour (wait for everything, give the memory back)
  if (!ctx->aux_streams.empty() || ctx->pool_bytes + bytes > POOL_MAX_BYTES || ctx->pool.size() >= POOL_MAX_BLOCKS) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(p);
    return;
  }
  ctx->pool.emplace_back(bytes, p);
  ctx->pool_bytes += bytes;
}
void* cfear_hpool_alloc(cfear_ctx* ctx, size_t bytes, size_t* got) {
  const size_t need = (bytes + 4095) & ~(size_t)4095;
  if (need > ((size_t)1 << 20)) return nullptr;  // mirrors are for the sweep-sized clouds of the per-call route
  for (size_t i = 0; i < ctx->hpool.size(); i++)
    if (ctx->hpool[i].first >= need && ctx->hpool[i].first <= 2 * need + 65536) {
      void* p = ctx->hpool[i].second; *got = ctx->hpool[i].first;
      ctx->hpool[i] = ctx->hpool.back(); ctx->hpool.pop_back();
      return p;
    }
  void* p = nullptr;
  if (hipHostMalloc(&p, need, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  *got = need;
  return p;
}
void cfear_hpool_free(cfear_ctx* ctx, void* p, size_t bytes) {
  if (!p) return;
  if (!ctx || ctx->hpool.size() >= 64) { (void)hipHostFree(p); return; }  // (stream-ordered reuse like the device pool: the next writer is a later kernel)
  ctx->hpool.emplace_back(bytes, p);
}
int cfear_ensure_hstage(cfear_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->h_stage_bytes) return CFEAR_OK;
  if (ctx->h_stage) { CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(ctx->h_stage); }
  ctx->h_stage = nullptr; ctx->h_stage_bytes = 0;
  const size_t want = (bytes + 65535) & ~(size_t)65535;
  if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), want, hipHostMallocDefault) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipHostMalloc staging");
  ctx->h_stage_bytes = want;
  return CFEAR_OK;
}

// One polar image (or a few) from host memory to the device on the context stream. Pinned memory (cfear_host_alloc) goes by DMA directly.
// Pageable memory - what a caller of radarDriver::CallbackOffline has (a cv::Mat, a std::vector) - is copied through a pinned staging
// buffer of the context in 256 KB pieces, each piece's DMA running while the next is copied: measured on MI355X, hipMemcpyAsync from a
// freshly allocated pageable 1.3 MB buffer took 600-700 us per sweep (2 GB/s; round 6, profiles/r06_dropin_phases.txt), the staged copy
// is bound by the host's memcpy. Larger transfers than the staging (many images at once) take the runtime's own path.
int cfear_upload_image(cfear_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  constexpr size_t STAGE_MAX = (size_t)64 << 20, PIECE = (size_t)256 << 10;
  hipPointerAttribute_t at;
  const bool pinned = hipPointerGetAttributes(&at, h_src) == hipSuccess && at.type == hipMemoryTypeHost;
  if (!pinned) (void)hipGetLastError();  // (an ordinary host pointer is reported as an error by some runtimes)
  if (pinned || bytes > STAGE_MAX) {
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return CFEAR_OK;
  }
  if (ctx->ev_img_pending) { CFEAR_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev_img)); ctx->ev_img_pending = false; }  // the previous image has left the staging
  if (bytes > ctx->h_img_bytes) {
    if (ctx->h_img) (void)hipHostFree(ctx->h_img);
    ctx->h_img = nullptr; ctx->h_img_bytes = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&ctx->h_img), bytes, hipHostMallocDefault) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipHostMalloc image staging");
    ctx->h_img_bytes = bytes;
  }
  if (!ctx->ev_img) CFEAR_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_img, hipEventDisableTiming));
  for (size_t off = 0; off < bytes; off += PIECE) {
    const size_t nb = bytes - off < PIECE ? bytes - off : PIECE;
    memcpy(ctx->h_img + off, static_cast<const unsigned char*>(h_src) + off, nb);
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(static_cast<unsigned char*>(d_dst) + off, ctx->h_img + off, nb, hipMemcpyHostToDevice, ctx->stream));
  }
  CFEAR_HIP_CHECK(ctx, hipEventRecord(ctx->ev_img, ctx->stream));
  ctx->ev_img_pending = true;
  return CFEAR_OK;
}

int cfear_ensure_staging(cfear_ctx* ctx, int n_scans) {
  const size_t pb = (size_t)n_scans * ctx->A * ctx->R + 64;
  const size_t sb = (size_t)n_scans * ctx->A * ctx->par.k_strongest * sizeof(uint32_t);
  if (pb > ctx->d_polar_bytes) {
    if (ctx->d_polar) (void)hipFree(ctx->d_polar);
    ctx->d_polar = nullptr; ctx->d_polar_bytes = 0;
    if (hipMalloc(&ctx->d_polar, pb) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc polar staging");
    ctx->d_polar_bytes = pb;
  }
  if (sb > ctx->d_slots_bytes) {
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    ctx->d_slots = nullptr; ctx->d_slots_bytes = 0;
    if (hipMalloc(&ctx->d_slots, sb) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc slot staging");
    ctx->d_slots_bytes = sb;
  }
  return CFEAR_OK;
}

int cfear_kstrongest_host(cfear_ctx* ctx, const uint8_t* h_polar, int n_scans, uint32_t* h_slots) {
  if (!ctx || !h_polar || !h_slots || n_scans <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "kstrongest_host: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = cfear_ensure_staging(ctx, n_scans);
  if (rc != CFEAR_OK) return rc;
  const size_t pb = (size_t)n_scans * ctx->A * ctx->R;
  const size_t sb = (size_t)n_scans * ctx->A * ctx->par.k_strongest * sizeof(uint32_t);
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_polar, h_polar, pb, hipMemcpyHostToDevice, ctx->stream));
  rc = cfear_launch_kstrongest(ctx, ctx->d_polar, n_scans, ctx->d_slots, ctx->stream);
  if (rc != CFEAR_OK) return rc;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(h_slots, ctx->d_slots, sb, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

int cfear_time_kstrongest(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, uint32_t* d_slots, int warmup,
                          int iters, double* avg_seconds) {
  if (!ctx || !avg_seconds || iters <= 0) return CFEAR_ERR_INVALID;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i < warmup; i++) {
    int rc = cfear_launch_kstrongest(ctx, d_polar, n_scans, d_slots, ctx->stream);
    if (rc != CFEAR_OK) return rc;
  }
  CFEAR_HIP_CHECK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  for (int i = 0; i < iters; i++) {
    int rc = cfear_launch_kstrongest(ctx, d_polar, n_scans, d_slots, ctx->stream);
    if (rc != CFEAR_OK) return rc;
  }
  CFEAR_HIP_CHECK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipEventSynchronize(ctx->ev1));
  float ms = 0.f;
  CFEAR_HIP_CHECK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  *avg_seconds = (double)ms * 1e-3 / iters;
  return CFEAR_OK;
}

}  // extern "C"
