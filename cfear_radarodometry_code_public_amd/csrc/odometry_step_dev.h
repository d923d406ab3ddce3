// odometry_step_dev.h -- one sweep of one sequence of the batched odometry (OdometryKeyframeFuser::processFrame,
// odometrykeyframefuser.cpp:143-259, split after the feature build :161): the device-side state, the kernel parameter block
// and the two step bodies, shared by pipeline.hip (one launch per stage and sweep) and replay.hip (a persistent workgroup per
// sequence that walks a whole chunk of sweeps).
#pragma once
#include "registration_dev.h"
#include "features_compact_dev.h"

using namespace cfear_dev;

namespace {

constexpr int BLOCK_F = CFEAR_FEAT_BLOCK;   // features / cloud kernels: two workgroups per compute unit (features_compact_dev.h)
constexpr int BLOCK_R = CFEAR_REG_BLOCK;   // registration kernels: 256 threads (4 waves = one per SIMD) in pipeline.hip
static_assert(BLOCK_R >= 64 * CFEAR_EVAL_WAVES, "the controller sums the partial results of CFEAR_EVAL_WAVES waves unconditionally");
constexpr int MAX_SCANS = 64;  // keyframes + current: the layout of SeqState in memory (the same in every translation unit, whatever CFEAR_REG_MAX_SCANS)
static_assert(CFEAR_REG_MAX_SCANS <= MAX_SCANS, "a registration cannot have more scans than a sequence keeps");
static_assert(FeatLdsC::total <= 80384, "two feature workgroups per compute unit");
struct RegLds {  // registration kernels
  static constexpr size_t red_d = 0;                                  // 10 sums x CFEAR_RED_STRIDE waves
  static constexpr size_t par = red_d + 10 * CFEAR_RED_STRIDE * sizeof(double);        // 3*CFEAR_REG_MAX_SCANS doubles
  static constexpr size_t red_i = par + 3 * CFEAR_REG_MAX_SCANS * sizeof(double);  // 64 ints
  static constexpr size_t scanptr = red_i + 64 * sizeof(int);        // CFEAR_REG_MAX_SCANS pointers
  static constexpr size_t regsh = scanptr + CFEAR_REG_MAX_SCANS * sizeof(void*);  // RegShared
  static constexpr size_t total = (regsh + sizeof(RegShared) + 15) / 16 * 16;
};

#ifndef CFEAR_REG_LDS_BUDGET
#define CFEAR_REG_LDS_BUDGET 53760  // three registration workgroups per compute unit (replay.hip: one, with the whole unit's LDS)
#endif
static_assert(RegLds::total + sizeof(double) * CFEAR_MATCH_LDS_DOUBLES <= CFEAR_REG_LDS_BUDGET,
              "registration kernels: more than 53,760 B of LDS costs the third workgroup per compute unit");

// Global per-sequence working memory (also one per context for the per-call API).
struct BlockScratch {
  uint64_t* keys; float* spts; int* order; int* vstart; int* vlist;  // big-cloud fallbacks of the LDS arrays
  int* vcur;        // [GRID_CAP + 2]
  int* rng;         // [cap_points][8] candidate row ranges per sample point
  double* part;     // [7][cap_points] partial cell moments per candidate chunk
  int* tmpi;        // [2 * cap_points + 16]
  float* samples;   // [cap_points * 3]
  double* match;    // [8][pair_cap]
  int* assoc;       // [3 * pair_cap]
  int cap_points, p2cap, pair_cap;
  const int* vrank; const int* vperm;  // FeatureScratch::vrank / vperm (cfear_scan_create under cfear_tune VOXEL_ORDER = 1; null otherwise)
};

struct SeqState {  // OdometryKeyframeFuser members (odometrykeyframefuser.h:203-260) for one sequence
  Aff2 T_prev, Tmot, Tcurrent;
  int nkf, free_slot, frames, last_slot;
  int ring[MAX_SCANS];
  Aff2 kf_pose[MAX_SCANS];
};

struct OdoParams {
  FeatureParams fp;
  RegParams rp;
  int A, k, compensate, ccw, use_keyframe, submap;
  double min_keyframe_dist, min_keyframe_rot_deg;
  long long* phase_times;  // optional [B][32] wall_clock64 ticks (tools/)
  long long* wg_times;     // optional [B][32]: start / end clock of every workgroup of the PRODUCTION kernels (slots 0, 1 features;
                           // 14, 15 registration) - two clock reads per workgroup, off the critical path
  int phase_detail;        // also accumulate the evaluation / controller times of every LM command (three clock reads per command:
                           // the registration kernel runs at half speed with them)
  int seq0;  // first sequence of this launch (sub-batches run on their own streams)
  // the scan slots of all sequences are one allocation, slot j of sequence q at scans_base + scan_stride * (q * (submap + 1) + j):
  // computed, not fetched from the pointer table (a memory round trip at the start of both kernels)
  unsigned char* scans_base; size_t scan_stride;
  cfear_sweep_record* records;  // optional [B]: this sweep's record of every sequence (cfear_odometry_replay_host)
  // registration workgroups longest first (cfear_tune REGISTRATION_ORDER): order[blockIdx.x] = the sequence a workgroup takes (null:
  // seq0 + blockIdx.x), work[q] = what sequence q's registration of THIS sweep cost (evaluations x residual blocks + associations),
  // the key the next sweep's order is sorted by (null: not recorded)
  const int* order; unsigned* work;
  int* flags;  // word 0 for the odometry object, word 1 + q for sequence q: bit 0 = some scan had more cells than its block holds, bit 1 = some cloud had more points
               // than the object is sized for (both CFEAR_ERR_CAPACITY); null: cannot happen
};

__device__ inline int next_pow2(int n) { int p = 1; while (p < n) p <<= 1; return p; }

// working memory of one feature build: global arrays of the general path (clouds beyond the compact path's limits) + the two
// LDS reduction arrays
__device__ __forceinline__ FeatureScratch make_fscratch(const BlockScratch& B, unsigned char* lds) {
  FeatureScratch W;
  W.keys = B.keys; W.spts = B.spts; W.order = B.order; W.vstart = B.vstart; W.vlist = B.vlist;
  W.vcur = B.vcur; W.lds = false; W.tab_voxels = 0;
  W.rng = B.rng; W.part = B.part; W.tmpi = B.tmpi;
  W.cap = B.cap_points;
  W.vrank = B.vrank; W.vperm = B.vperm;
  W.samples = B.samples;
  W.red_i = reinterpret_cast<int*>(lds + FeatLdsC::red_i);
  W.red_f = reinterpret_cast<float*>(lds + FeatLdsC::red_f);
  return W;
}
// cfear_tune NN_TIE_RULE = 2: one thread builds the scan's kd-tree (kdtree_flann_dev.h) over the float cell means; the activation stack of the
// build lives in the partial-moment scratch (free once the cells exist). A scan without the arrays (created before the mode was switched on)
// or a stack that does not suffice is reported through the scan's status.
__device__ inline void kd_build_block(ScanDev* S, const BlockScratch& B) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int frames = (int)(((size_t)7 * B.cap_points * sizeof(double)) / sizeof(KdFrame));
    const bool ok = S->kd.nodes && kd_build_serial(&S->kd, S->mean_f, S->n_cells, reinterpret_cast<KdFrame*>(B.part), frames);
    if (!ok && S->status == 0) S->status = CFEAR_ERR_UNSUPPORTED;
  }
  __syncthreads();
}
// byte_intensities: every intensity of the cloud is an integer in 0..255 (block-uniform; always true for clouds made from
// the filter's slots) - the compact path keeps them as bytes
// (the batched cloud pass leaves the cloud in registers only: the general path needs it in memory and writes it first)
__device__ __forceinline__ void features_dispatch(ScanDev* S, int n, const FeatureParams& P, const BlockScratch& B,
                                                  unsigned char* lds, PhaseTimer* pt, const float* bounds, bool zeroed, bool byte_intensities,
                                                  const PointRegs& PR, bool registers_only = false) {
  const FeatureScratch W = make_fscratch(B, lds);
  if (!(byte_intensities && !W.vrank && features_block_c(S, n, P, W, lds, pt, bounds, zeroed, PR))) {  // (a given voxel order: the general path)
    // registers_only (a constant of the call site): the cloud pass may have left the cloud in registers only (byte intensities)
    if (registers_only && PR.rounds > 0) point_regs_to_global(PR, S->xyi);  // block-uniform
    features_block(S, n, P, W, next_pow2(n), pt, bounds);
  }
  if (P.nn_tie == 2) kd_build_block(S, B);  // parity mode: the kd-tree FLANN builds over the cell means (ComputeSearchTreeFromCells, pointnormal.cpp:151-162)
}
__device__ inline RegScratch make_rscratch(const BlockScratch& B, unsigned char* lds) {
  RegScratch W;
  const size_t c = (size_t)B.pair_cap;
  W.tmx = B.match; W.tmy = B.match + c; W.a0 = B.match + 2 * c; W.a1 = B.match + 3 * c; W.a2 = B.match + 4 * c;
  W.sx = B.match + 5 * c; W.sy = B.match + 6 * c; W.w = B.match + 7 * c;
  W.assoc = B.assoc; W.cap = B.pair_cap; W.acap = 3 * B.pair_cap;  // (scratch_layout: three ints per pair)
  W.red = reinterpret_cast<double*>(lds + RegLds::red_d);
  W.red_i = reinterpret_cast<int*>(lds + RegLds::red_i);
  return W;
}

// ---- batched odometry: OdometryKeyframeFuser::processFrame (odometrykeyframefuser.cpp:143-259) with all
// state on the device, split after the feature build (:161) ------------------------------------------
// TIMED: per-phase timestamps (tools/); the production instantiation carries no timer at all
template <bool TIMED>
__device__ __forceinline__ void features_step_body(unsigned char* lds /* FeatLdsC::total bytes */, int q, const uint32_t* slots_all, const double* trig,
                                                   const OdoParams& OP, const SeqState* states, const BlockScratch* scratch) {
  const SeqState* st = &states[q];
  const BlockScratch B = scratch[q];
  ScanDev* cur = reinterpret_cast<ScanDev*>(OP.scans_base + OP.scan_stride * ((size_t)q * (OP.submap + 1) + st->free_slot));
  const Aff2 TprevMot = st->Tmot;  // :146
  PhaseTimer pt; pt.t = (TIMED && OP.phase_times) ? OP.phase_times + (size_t)q * 32 : nullptr; pt.n = 0; pt.cap = 14; pt.acc = nullptr;
  if (TIMED) pt.mark();
  if (!TIMED && OP.wg_times && threadIdx.x == 0) OP.wg_times[(size_t)q * 32] = (long long)wall_clock64();
  // the voxel bitmap starts all-zero (cleared here: the barriers of the cloud pass publish it)
  {
    uint32_t* z = reinterpret_cast<uint32_t*>(lds + FeatLdsC::bm);
    for (int i = threadIdx.x; i < (int)((FeatLdsC::bmp - FeatLdsC::bm) / 4); i += BLOCK_F) z[i] = 0u;
  }
  // stage 1 (second half) + 1.5: slots -> cloud (radar_driver.cpp:59), motion compensation (:147-150), bounding box
  double mot[3]; aff_to_xyt(TprevMot, mot);
  float bounds[4];
  PointRegs PR;
  // (the compact feature path works from the registers: the 58 KB of the cloud are only written when somebody reads them)
  const int n = cloud_step_block(slots_all + (size_t)q * OP.A * OP.k, OP.A, OP.k, trig, OP.fp.range_res, OP.fp.min_distance,
                                 cur->xyi, cur->cap_points, OP.compensate, mot[0], mot[1], mot[2], OP.ccw,
                                 reinterpret_cast<int*>(lds + FeatLdsC::red_i), reinterpret_cast<float*>(lds + FeatLdsC::red_f),
                                 reinterpret_cast<double*>(lds + FeatLdsC::pxy),  // 6 doubles per bearing where the sorted points go later
                                 (int)(CFEAR_CPT_CAP * 8 / (6 * sizeof(double))), bounds, PR, false);
  if (TIMED) { pt.mark(); pt.mark(); }
  CFEAR_STOP_AT(1, );
  features_dispatch(cur, n, OP.fp, B, lds, TIMED ? &pt : nullptr, n > 0 ? bounds : nullptr, true, true, PR, true);  // :161
  if (OP.flags && threadIdx.x == 0 && cur->status == CFEAR_ERR_CAPACITY) { atomicOr(OP.flags, 1); OP.flags[1 + q] |= 1; }  // (thread 0 wrote the status itself; word 1 + q: this sequence's own)
  if (!TIMED && OP.wg_times && threadIdx.x == 0) OP.wg_times[(size_t)q * 32 + 1] = (long long)wall_clock64();
}

// The same stage from a CLOUD on the device (filter_type CA-CFAR, radar_driver.cpp:52-56, or cfear_odometry_step_cloud_device): sequence
// q's points are xyi_all + q * cap * 3, min(counts[q], cap) of them. Compensate (odometrykeyframefuser.cpp:146-150, utils.cpp:96-113:
// atan2 / sin / cos per point as written - a detector's points do not sit on bearing rays the per-bearing table of cloud_step_block
// could expand around) and MapPointNormal (:161) through the same dispatch as cfear_scan_create: the compact path when the cloud fits
// it (byte intensities, <= 4864 points), the general path in the sequence's global arrays otherwise.
__device__ __forceinline__ void features_cloud_step_body(unsigned char* lds /* FeatLdsC::total bytes */, int q, const float* xyi_all, int cap, const int* counts,
                                                         const OdoParams& OP, const SeqState* states, const BlockScratch* scratch) {
  const SeqState* st = &states[q];
  const BlockScratch B = scratch[q];
  ScanDev* cur = reinterpret_cast<ScanDev*>(OP.scans_base + OP.scan_stride * ((size_t)q * (OP.submap + 1) + st->free_slot));
  const Aff2 TprevMot = st->Tmot;  // :146
  if (OP.wg_times && threadIdx.x == 0) OP.wg_times[(size_t)q * 32] = (long long)wall_clock64();
  int n = counts[q];
  bool clipped = false;
  if (n > cap) { n = cap; clipped = true; }
  if (n > cur->cap_points) { n = cur->cap_points; clipped = true; }
  if (n < 0) n = 0;
  if (clipped && OP.flags && threadIdx.x == 0) { atomicOr(OP.flags, 2); OP.flags[1 + q] |= 2; }  // more detections than the object is sized for (cfar_max_points)
  const float* src = xyi_all + 3 * (size_t)q * cap;
  int bytes = 1;
  for (int i = threadIdx.x; i < n; i += BLOCK_F) {
    const float x = src[3 * i], y = src[3 * i + 1], w = src[3 * i + 2];
    cur->xyi[3 * i] = x; cur->xyi[3 * i + 1] = y; cur->xyi[3 * i + 2] = w;
    bytes &= (w >= 0.f && w <= 255.f && w == (float)(int)w) ? 1 : 0;
  }
  const bool byte_intensities = __syncthreads_and(bytes) != 0;  // (the barrier also makes the copy visible to the whole block)
  if (OP.compensate) {
    double mot[3]; aff_to_xyt(TprevMot, mot);
    compensate_block(cur->xyi, n, mot[0], mot[1], mot[2], OP.ccw);  // :147 (the peaks cloud of :149 is empty and never read)
  }
  PointRegs PR;
  point_regs_from_global(cur->xyi, n, PR);
  features_dispatch(cur, n, OP.fp, B, lds, nullptr, nullptr, false, byte_intensities, PR);  // :161
  if (OP.flags && threadIdx.x == 0 && cur->status == CFEAR_ERR_CAPACITY) { atomicOr(OP.flags, 1); OP.flags[1 + q] |= 1; }
  if (OP.wg_times && threadIdx.x == 0) OP.wg_times[(size_t)q * 32 + 1] = (long long)wall_clock64();
}

template <bool TIMED, int KCOST = -1>
__device__ __forceinline__ void register_step_body(unsigned char* lds /* RegLds::total bytes */, int q, const OdoParams& OP, SeqState* states,
                                                   const BlockScratch* scratch, double* cov_work /*[B][36]*/, cfear_reg_summary* summaries,
                                                   double* poses_out /*[B][3]*/) {
  const int tid = threadIdx.x;
  SeqState* st = &states[q];
  const BlockScratch B = scratch[q];
  const int nslots = OP.submap + 1;
  auto slot_ptr = [&](int j) -> ScanDev* { return reinterpret_cast<ScanDev*>(OP.scans_base + OP.scan_stride * ((size_t)q * nslots + j)); };
  const int cur_slot = st->free_slot;
  ScanDev* cur = slot_ptr(cur_slot);
  const Aff2 T_prev = st->T_prev, TprevMot = st->Tmot;
  const int nkf = st->nkf;
  // this thread's keyframe (threads below nkf), read with the rest of the state: one round trip, before the barrier
  const int my_ring = st->ring[tid < MAX_SCANS ? tid : 0];
  const Aff2 my_kf_pose = st->kf_pose[tid < MAX_SCANS ? tid : 0];
  PhaseTimer pt; pt.t = (TIMED && OP.phase_times) ? OP.phase_times + (size_t)q * 32 + 14 : nullptr; pt.n = 0; pt.cap = 15;
  pt.acc = (TIMED && OP.phase_times && OP.phase_detail) ? OP.phase_times + (size_t)q * 32 + 29 : nullptr;
  pt.acc2 = (TIMED && OP.phase_times && OP.phase_detail == 2) ? OP.phase_times + (size_t)q * 32 : nullptr;  // (over the feature kernel's stamps)
  if (TIMED && pt.acc2 && tid == 0) for (int i = 0; i < 8; i++) pt.acc2[i] = 0;
  if (TIMED) pt.mark();
  if (!TIMED && OP.wg_times && tid == 0) OP.wg_times[(size_t)q * 32 + 14] = (long long)wall_clock64();
  const Aff2 Tguess = aff_mul(T_prev, TprevMot);  // :166
  cfear_reg_summary* sum = &summaries[q];
  __syncthreads();  // every thread has read the state before thread 0 rewrites it
  if (nkf == 0) {  // :171-177
    if (tid == 0) {
      st->ring[0] = cur_slot; st->kf_pose[0] = aff_identity(); st->nkf = 1; st->free_slot = (cur_slot + 1) % nslots;
      st->frames++; st->last_slot = cur_slot;
      if (OP.work) OP.work[q] = 0u;
      sum->success = 0; sum->usable = 0; sum->outer_iterations = 0; sum->num_residuals = 0; sum->num_residual_blocks = 0;
      double v[3]; aff_to_xyt(st->Tcurrent, v);
      poses_out[3 * q] = v[0]; poses_out[3 * q + 1] = v[1]; poses_out[3 * q + 2] = v[2];
      if (OP.records) {
        cfear_sweep_record* r = OP.records + q;
        r->pose[0] = v[0]; r->pose[1] = v[1]; r->pose[2] = v[2]; r->final_cost = 0;
        r->outer_iterations = 0; r->num_residuals = 0; r->n_keyframes = 1; r->n_cells = cur->n_cells;
        for (int i = 0; i < 8; i++) r->inner_iterations[i] = 0;
      }
    }
    return;
  }
  // FormatScans (:478-494)
  ScanDev** sp = reinterpret_cast<ScanDev**>(lds + RegLds::scanptr);
  double* poses = reinterpret_cast<double*>(lds + RegLds::par);  // the registration's parameter array itself: the poses never go through memory
  const int ns = nkf + 1;
  if (tid < nkf) {
    sp[tid] = slot_ptr(my_ring);
    double v[3]; aff_to_xyt(my_kf_pose, v);
    poses[3 * tid] = v[0]; poses[3 * tid + 1] = v[1]; poses[3 * tid + 2] = v[2];
  }
  if (tid == 0) {
    sp[ns - 1] = cur;
    double v[3]; aff_to_xyt(Tguess, v);
    poses[3 * (ns - 1)] = v[0]; poses[3 * (ns - 1) + 1] = v[1]; poses[3 * (ns - 1) + 2] = v[2];
  }
  // cov_vek.back() as FormatScans leaves it (Identity66, :486-490): what cov_current is when the registration has no usable
  // solution (it is in/out for register_block); read back through cfear_odometry_covariances
  if (tid >= 64 && tid < 100) cov_work[(size_t)q * 36 + (tid - 64)] = ((tid - 64) % 7 == 0) ? 1.0 : 0.0;
  __syncthreads();
  const RegScratch RW = make_rscratch(B, lds);
  register_block<KCOST>(sp, ns, poses, cov_work + (size_t)q * 36, OP.rp, RW, reinterpret_cast<double*>(lds + RegLds::par),
                 reinterpret_cast<RegShared*>(lds + RegLds::regsh), sum, TIMED ? &pt : nullptr);  // :186 (result ignored, :184-186)
  __syncthreads();
  if (TIMED) pt.mark();
  if (tid == 0) {
    Aff2 Tcurrent = aff_from_xyt(poses[3 * (ns - 1)], poses[3 * (ns - 1) + 1], poses[3 * (ns - 1) + 2]);  // :195
    const Aff2 Tpi = aff_inv(T_prev);
    const Aff2 Tmot_current = aff_mul(Tpi, Tcurrent);
    {  // AccelerationVelocitySanityCheck (:76-94)
      const double dt = 0.25, lim = 200;
      const double vel = sqrt(Tmot_current.t0 * Tmot_current.t0 + Tmot_current.t1 * Tmot_current.t1) / dt;
      const double ax = (Tmot_current.t0 - TprevMot.t0) / (dt * dt), ay = (Tmot_current.t1 - TprevMot.t1) / (dt * dt);
      const double acc = sqrt(ax * ax + ay * ay);
      if (acc > lim || vel > lim) Tcurrent = Tguess;  // :198-199
    }
    st->Tmot = aff_mul(Tpi, Tcurrent);  // :200
    st->Tcurrent = Tcurrent;
    const Aff2 Tkeydiff = aff_mul(aff_inv(st->kf_pose[nkf - 1]), Tcurrent);  // :227
    bool fuse = true;
    if (OP.use_keyframe) {  // KeyFrameBasedFuse (:62-73)
      const double tn = sqrt(Tkeydiff.t0 * Tkeydiff.t0 + Tkeydiff.t1 * Tkeydiff.t1);
      const double rot = fabs(atan2(Tkeydiff.l2, Tkeydiff.l3));
      fuse = (tn > OP.min_keyframe_dist) || (rot > OP.min_keyframe_rot_deg * 3.14159265358979323846 / 180.0);
    }
    if (fuse) {  // AddToReference (:470-476)
      int m = nkf;
      st->ring[m] = cur_slot; st->kf_pose[m] = Tcurrent; m++;
      int freed;
      if (m > OP.submap) {
        freed = st->ring[0];
        for (int i = 0; i + 1 < m; i++) { st->ring[i] = st->ring[i + 1]; st->kf_pose[i] = st->kf_pose[i + 1]; }
        m--;
      } else {
        freed = m;  // slots are handed out in increasing order until the ring is full
      }
      st->nkf = m; st->free_slot = freed;
    }
    st->T_prev = Tcurrent;  // :257
    st->frames++; st->last_slot = cur_slot;
    double v[3]; aff_to_xyt(Tcurrent, v);
    poses_out[3 * q] = v[0]; poses_out[3 * q + 1] = v[1]; poses_out[3 * q + 2] = v[2];
    if (OP.records) {  // what a caller of pointcloudCallback sees after this sweep, kept per sweep (no host round trip in a replay)
      cfear_sweep_record* r = OP.records + q;
      r->pose[0] = v[0]; r->pose[1] = v[1]; r->pose[2] = v[2]; r->final_cost = sum->final_cost;
      r->outer_iterations = sum->outer_iterations; r->num_residuals = sum->num_residuals; r->n_keyframes = st->nkf; r->n_cells = cur->n_cells;
      for (int i = 0; i < 8; i++) r->inner_iterations[i] = sum->inner_iterations[i];
    }
    if (OP.work) {  // from the solver state in LDS: no read-back of the summary from memory
      const RegShared* rs = reinterpret_cast<const RegShared*>(lds + RegLds::regsh);
      int evals = 0;
      for (int i = 0; i < rs->nrec; i++) evals += rs->orec[i].inner;
      const int nsrc = rs->kf[ns - 1].n_cells;
      OP.work[q] = (unsigned)evals * (unsigned)rs->M + 6u * (unsigned)rs->nrec * (unsigned)(nkf * nsrc);
    }
    if (!TIMED && OP.wg_times) OP.wg_times[(size_t)q * 32 + 15] = (long long)wall_clock64();
  }
}



#ifndef CFEAR_REG_MIN_WG
#define CFEAR_REG_MIN_WG 3  // workgroups per compute unit the registration step kernel is compiled for (tools: A/B builds)
#endif
// KCOST: the cost metric the kernel is compiled for (registration_dev.h evaluate_partial), -1: any (the timed instantiation)
#ifdef CFEAR_REG_NUM_VGPR  // tools/reg_resources.sh: cap the architectural VGPRs below the occupancy budget (the rest of it: AGPRs for spills)
#define CFEAR_REG_VGPR_ATTR __attribute__((amdgpu_num_vgpr(CFEAR_REG_NUM_VGPR)))
#else
#define CFEAR_REG_VGPR_ATTR
#endif
template <bool TIMED, int KCOST>
__global__ __launch_bounds__(BLOCK_R, CFEAR_REG_MIN_WG) CFEAR_REG_VGPR_ATTR void register_step_kernel(OdoParams OP, SeqState* states, ScanDev* const* scan_slots,
                                                                const BlockScratch* scratch, double* poses_work /*[B][MAX_SCANS*3]*/,
                                                                double* cov_work /*[B][36]*/, cfear_reg_summary* summaries,
                                                                double* poses_out /*[B][3]*/) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[RegLds::total];
  register_step_body<TIMED, KCOST>(lds, OP.order ? OP.order[blockIdx.x] : OP.seq0 + (int)blockIdx.x, OP, states, scratch, cov_work, summaries, poses_out);
}


}  // namespace
