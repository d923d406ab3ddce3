// common.h -- shared host-side definitions of libcfear_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/cfear_hip.h"

#define CFEAR_WAVE 64

struct cfear_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  cfear_params par;
  int A = 0, R = 0;
  std::string err;
  // per-azimuth (cos, sin) of theta=(b+1)/A*2pi computed on the host with libm so that the
  // polar->Cartesian conversion (radar_filters.cpp:317-330) is bit-identical to a CPU run.
  double* d_trig = nullptr;  // [A][2]
  // staging for the host entry points
  uint8_t* d_polar = nullptr;
  size_t d_polar_bytes = 0;
  uint32_t* d_slots = nullptr;
  size_t d_slots_bytes = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // scratch for the per-call feature / registration kernels
  void* d_scratch = nullptr;
  size_t scratch_bytes = 0;
  int* d_cfar_rows = nullptr;  // row counts / bases of cfear_filter_cfar_batch_device
  size_t cfar_rows_cap = 0;
  // streams of the batched odometry objects of this context (cfear_synchronize waits for them too)
  std::vector<hipStream_t> aux_streams;
  // launch-shape knobs (cfear_tune): filter occupancy variant (5..7 waves per SIMD), rows walked per filter wave,
  // whether a batched odometry object created from now on runs its filter one sweep ahead on a stream of its own
  int tune_k1_occ = 7, tune_k1_rows = 0 /* 0 = by launch size: 4, or 6 from 1536 scans up */, tune_odo_overlap = 0;
  int tune_filter_cus = 0;  // with ODOMETRY_OVERLAP: compute units reserved for the filter stream (CU-masked streams); 0 = no masks
  int tune_reg_order = 1;  // batched odometry objects created afterwards launch their registration workgroups longest first (keys: the previous sweep's work)
  int tune_max_cells = 0;  // batched odometry: oriented surface points per scan the scan blocks / match scratch are sized for (0 = A * k: every filtered point)
  int tune_large_kernel = 0;  // batched odometry, submap_scan_size > 7: 0 = the 512-thread kernel of register_step_large.hip when the sequences fit the chip at one per compute unit, 1 = never, 2 = always
  int tune_voxel_order = 0;  // per-call scans: 0 = a voxel's points summed in index order (production: the stable order), 1 = in the order libstdc++'s std::sort leaves them (PCL <= 1.9)
  int tune_nn_tie = 0;  // which of several exactly equidistant cells GetClosestIdx returns: 0 lowest index (production), 1 highest, 2 FLANN's kd-tree order (parity mode, slow)
  int tune_repeat_shortcut = 1;  // registration: an outer iteration that would repeat the previous one bit for bit is not recomputed (0: it is - tests)
  int tune_replay_persistent_max = 256;  // cfear_odometry_replay_host: up to this many sequences run as persistent workgroups (replay.hip)
  // Device blocks handed back by cfear_cloud_release / cfear_scan_release, reused by the next allocation of a similar size (round 6): the
  // per-call route (radarDriver -> Compensate -> MapPointNormal -> Register, include/cfear_hip/cfear_host.hpp) creates and drops two clouds
  // and a scan per sweep, and hipMalloc / hipFree (the latter a device-wide synchronisation) cost more than its kernels. Reuse is
  // stream-ordered: every per-call entry point works on ctx->stream (a context with odometry objects on streams of their own keeps the
  // synchronising release).
  std::vector<std::pair<size_t, void*>> pool;
  size_t pool_bytes = 0;
  std::vector<std::pair<size_t, void*>> hpool;  // ... and the pinned host blocks of the clouds' mirrors
  // pinned host staging of the per-call entry points: downloads that complete with ONE synchronisation, registration arguments in one copy
  unsigned char* h_stage = nullptr;
  size_t h_stage_bytes = 0;
  // pinned staging of one polar image for callers that hand over pageable memory (cfear_upload_image), and the event after its last copy
  unsigned char* h_img = nullptr;
  size_t h_img_bytes = 0;
  hipEvent_t ev_img = nullptr;
  bool ev_img_pending = false;
};

static inline int cfear_fail(cfear_ctx* c, int code, const char* what, hipError_t e = hipSuccess) {
  if (c) {
    char buf[512];
    if (e != hipSuccess)
      snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    else
      snprintf(buf, sizeof(buf), "%s", what);
    c->err = buf;
  }
  return code;
}

#define CFEAR_HIP_CHECK(ctx, call)                                            \
  do {                                                                        \
    hipError_t _e = (call);                                                   \
    if (_e != hipSuccess) return cfear_fail((ctx), CFEAR_ERR_HIP, #call, _e); \
  } while (0)

struct cfear_cloud {  // pcl::PointCloud<pcl::PointXYZI> on the device: one block, the count in its first 16 bytes
  int cap = 0;
  float* d_xyi = nullptr;  // [cap][3] x, y, intensity (= block + 16)
  int* d_n = nullptr;      // point count (= block)
  void* block = nullptr;
  size_t bytes = 0;
  // Host mirror (round 6): a pinned, device-visible copy of the block that the kernels producing or changing the cloud on the per-call route write as
  // well - cfear_clouds_download then is a wait and a memcpy, no copy command (kernel + D2H + wait 21 us, kernel writing host memory + wait 13 us:
  // profiles/r06_sync_latency.txt; the per-call route downloads four clouds a sweep). mirror_valid: the mirror holds what the device block holds.
  unsigned char* h_block = nullptr;
  size_t h_bytes = 0;
  bool mirror_valid = false;
  int* h_n() const { return reinterpret_cast<int*>(h_block); }
  float* h_xyi() const { return reinterpret_cast<float*>(h_block + 16); }
};

// cabi.hip
extern "C" __attribute__((visibility("hidden"))) int cfear_ensure_staging(cfear_ctx* ctx, int n_scans);
extern "C" __attribute__((visibility("hidden"))) int cfear_pool_alloc(cfear_ctx* ctx, size_t bytes, void** out, size_t* got);
extern "C" __attribute__((visibility("hidden"))) void cfear_pool_free(cfear_ctx* ctx, void* p, size_t bytes);
extern "C" __attribute__((visibility("hidden"))) int cfear_ensure_hstage(cfear_ctx* ctx, size_t bytes);
extern "C" __attribute__((visibility("hidden"))) void* cfear_hpool_alloc(cfear_ctx* ctx, size_t bytes, size_t* got);  // null: no mirror (not an error)
extern "C" __attribute__((visibility("hidden"))) void cfear_hpool_free(cfear_ctx* ctx, void* p, size_t bytes);
extern "C" __attribute__((visibility("hidden"))) int cfear_upload_image(cfear_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
// pipeline.hip
extern "C" __attribute__((visibility("hidden"))) int cfear_cloud_alloc(cfear_ctx* ctx, int cap, cfear_cloud** out);
// kstrongest.hip
__attribute__((visibility("hidden"))) int cfear_launch_kstrongest(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, uint32_t* d_slots, hipStream_t stream);

// cfar.hip
__attribute__((visibility("hidden"))) int cfear_launch_cfar_batch(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, int window_size, int nb_guard_cells,
                                                                  float false_alarm_rate, double max_distance, float* d_xyi, int capacity, int* d_counts,
                                                                  int* d_rows /* cfear_cfar_scratch_ints(ctx, n_scans) ints */, hipStream_t stream);
__attribute__((visibility("hidden"))) size_t cfear_cfar_scratch_ints(const cfear_ctx* ctx, size_t n_scans);

// scans (keyframes + current) the batched registration kernels of register_step.hip are compiled for: pipeline.hip launches them
// when submap_scan_size + 1 fits, its own 64-scan instantiation otherwise
#define CFEAR_STEP_SMALL_SCANS 8
