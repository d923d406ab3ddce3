// features_dev.h -- K3/K4/K5 device code: polar slots -> Cartesian cloud (+ motion compensation)
// -> oriented surface points (MapPointNormal) + the uniform grid that replaces the FLANN kd-tree
// over cell means. One workgroup builds one scan; all functions are block-collective.
//
// Reference: radar_filters.cpp:309-337 (cloud), utils.cpp:96-113 + utils.h:28-32 (compensation),
// pointnormal.cpp:7-63, 151-162, 265-297 (features). Third-party behaviour restated from
// SURVEY.md section 9 (PCL VoxelGrid, FLANN radius search).
#pragma once
#include <type_traits>
#include "blockops.h"
#include "kdtree_flann_dev.h"
#include "../../include/cfear_hip.h"

namespace cfear_dev {

#define CFEAR_HAVE_KD 1
// Device-resident MapPointNormal. Lives in device memory; the arrays are one flat allocation.
struct ScanDev {
  int n_points;   // input_ size (cloud after the min-range cut, compensated)
  int n_samples;  // voxel centroids
  int n_cells;    // valid cells
  int status;     // 0 ok, CFEAR_ERR_EMPTY on an empty cloud
  float gminx, gminy, gcell;  // uniform grid over float cell means
  int gw, gh;
  int cap_points, cap_cells, cap_grid;
  float* xyi;         // [cap_points][3]
  cfear_cell* cells;  // [cap_cells], or null: the scans of a batched odometry object do not keep the 120-byte records - nothing on the path reads
                      // them (the registration reads rsrc / rtar / rcov, the search gpts): 24 KB per scan that were written for nobody
  float* mean_f;      // [cap_cells][2]  (downsampled_, pointnormal.cpp:151-158)
  int* gstart;        // [cap_grid + 4] 32-bit bucket offsets (scans of more than CFEAR_GRID16_MAX cells), followed by the 16-bit offsets (grid_off16)
  float4* gpts;       // [cap_cells] (mean x, mean y, cell index bits, 0) in bucket order: the 1-NN scan reads contiguously
  // registration views of the cells (the association is bound by the number of scattered load instructions,
  // so the fields it needs are packed): mean x, mean y, normal x, normal y, nsamples, scale
  double* rsrc;       // [6][cap_cells] SoA: read with consecutive cell indices when the scan is the source
  double* rtar;       // [cap_cells][8] 64-byte records: read at random cell indices when the scan is a target
  double* rcov;       // [cap_cells][3] covariance xx, xy, yy: what the P2D cost reads of a target cell besides its rtar record (n_scan_normal.cpp:290-299)
  KdTree kd;          // parity mode (cfear_tune NN_TIE_RULE = 2): the kd-tree FLANN would build over mean_f (kdtree_flann_dev.h); arrays null otherwise
};
#define CFEAR_GRID_CAP (128 * 128)  // buckets per scan (ScanDev::cap_grid)
// Scans of up to CFEAR_GRID16_MAX cells (every scan the odometry builds) keep their bucket offsets as 16-bit values in the
// region behind gstart: off16[g] = cells in buckets < g for g = 0 .. G, followed by 2 * gw more entries equal to off16[G] (the
// 1-NN search reads the offsets of bucket g, g + gw and g + 2 gw - three rows of its window - without clamping). A scan of
// ~200 cells spreads over ~10 000 buckets: 20 KB instead of the 120 KB of 32-bit offsets plus 8-byte three-row records.
// Bigger scans (per-call API only) keep 32-bit offsets in gstart itself.
#define CFEAR_GRID16_MAX 0xFFF0
__device__ __forceinline__ unsigned short* grid_off16(int* gstart) { return reinterpret_cast<unsigned short*>(gstart + CFEAR_GRID_CAP + 4); }
__device__ __forceinline__ const unsigned short* grid_off16(const int* gstart) { return reinterpret_cast<const unsigned short*>(gstart + CFEAR_GRID_CAP + 4); }

struct FeatureParams {
  int nn_tie;                // cfear_tune NN_TIE_RULE: 2 = also build the FLANN kd-tree over the cell means (parity mode)
  float range_res, min_distance;
  float radius;              // (float)par.res, pointnormal.h:118
  double downsample_factor;  // pointnormal.h:241
  int weight_intensity;
  double assoc_radius;       // registration.h:122 (sizes the NN grid)
};

// Instruction profile per phase (tools/build_stop_variants.sh, never the product build): -DCFEAR_FEATURES_STOP=k makes the
// feature build return after phase k, so that the PMC counters of successive variants difference into per-phase counts.
#ifdef CFEAR_FEATURES_STOP
#define CFEAR_STOP_AT(k, ret) do { if (CFEAR_FEATURES_STOP == (k)) return ret; } while (0)
#else
#define CFEAR_STOP_AT(k, ret) do { } while (0)
#endif

// optional per-block phase timestamps (bring-up / tuning): thread 0 stores wall_clock64() ticks (100 MHz)
struct PhaseTimer {
  long long* t;    // next free tick slot of this kernel's share of the 32 per sequence
  int n, cap;
  long long* acc;  // three accumulators (slots 29..31 of the sequence) or null
  long long* acc2 = nullptr;  // eight accumulators of the controller breakdown (phase_detail 2; slots 0..7 of the sequence) or null
  __device__ inline void mark() { if (t && threadIdx.x == 0 && n < cap) t[n++] = (long long)wall_clock64(); }
};

// Working memory of one block in global memory: the general feature path (features_block below; clouds of any size) works
// in it, the compact path (features_compact_dev.h: two workgroups per compute unit, the one every reference configuration
// takes) only uses the partial sums, the reduction arrays and - for the cell grid - an LDS region handed over in `keys`.
struct FeatureScratch {
  uint64_t* keys;   // [p2] (voxel << 32 | point) sort keys. NB: a 24-bit packing with an '& 0xFFFFFF' extract is
                    // miscompiled by hipcc 7.2 (mask dropped in front of v_mad_u64_u32)
  float* spts;      // [n][3] points in sorted order
  int* order;       // [n] point index of every sorted position
  int* vstart;      // [n + 1] start of each occupied voxel in the sorted order
  int* vlist;       // [n] voxel index of each occupied voxel, ascending
  int* vcur;        // global [cap_grid + 1] cursors of the cell-mean grid
  float* samples;   // global [cap_points][3] voxel centroids
  int* red_i;       // LDS, >= 64 ints
  float* red_f;     // LDS, >= 64 floats
  bool lds;         // cell_grid_block only: `keys` is an LDS region of tab_voxels / 2 ints for its counters, `vlist` the float cell means
  int* rng;         // global [cap_points][8] candidate row ranges of every sample point
  double* part;     // global [7][cap_points] partial moments of the candidate chunks
  int* tmpi;        // global [2 * cap + 16] copies of vlist/vstart (only used when leaf < radius)
  const int* vrank; const int* vperm;  // cfear_tune VOXEL_ORDER = 1 (per-call scans, general path): rank of every point in the order PCL <= 1.9's
                    // std::sort leaves the VoxelGrid's points in, and the point at every rank; null: points of a voxel in index order
  int cap;          // capacity (entries) of order/vstart/vlist/rng/part
  int tab_voxels;   // cell_grid_block only: twice the ints the LDS region in `keys` holds
};

#define CFEAR_TWO_PI 6.283185307179586476925286766559
// threads of every workgroup that runs the code of this header and of features_compact_dev.h (pipeline.hip / replay.hip BLOCK_F): a
// compile-time constant - blockDim.x is a load from the dispatch packet, a division by it an integer division at run time, and a
// block-wide scan over "blockDim.x / 64" partial sums a loop instead of eight reads side by side
#define CFEAR_FEAT_BLOCK 512
#define CFEAR_INV_TWO_PI 0.15915494309189533576888376337251

// getPeaksFilteredPointCloud (radar_filters.cpp:309-337): row-major over (bearing, slot).
// trig[b] = (cos, sin) of theta = (b+1)/A*2pi computed by the host libm. Returns the point count.
__device__ inline int cloud_build_block(const uint32_t* __restrict__ slots, int A, int k,
                                        const double* __restrict__ trig, float range_res_f, float min_distance_f,
                                        int peaks, float* __restrict__ xyi, int cap, int* red_i) {
  const double range_res = (double)range_res_f;
  const int min_range_bin = (int)ceil((double)min_distance_f / range_res);  // :315
  const double range_res_half = range_res / 2.0;
  const int items = A * k;
  const int ipt = (items + CFEAR_FEAT_BLOCK - 1) / CFEAR_FEAT_BLOCK;
  const int i0 = threadIdx.x * ipt, i1 = min(items, i0 + ipt);
  int cnt = 0;
  for (int i = i0; i < i1; i++) {
    const uint32_t s = slots[i];
    const bool ok = CFEAR_SLOT_VALID(s) && (!peaks || CFEAR_SLOT_PEAK(s)) && CFEAR_SLOT_RANGE(s) > min_range_bin;  // :327
    cnt += ok ? 1 : 0;
  }
  int total;
  int o = block_exclusive_scan<CFEAR_FEAT_BLOCK>(cnt, red_i, &total);
  for (int i = i0; i < i1; i++) {
    const uint32_t s = slots[i];
    const int range = CFEAR_SLOT_RANGE(s);
    const bool ok = CFEAR_SLOT_VALID(s) && (!peaks || CFEAR_SLOT_PEAK(s)) && range > min_range_bin;
    if (ok && o < cap) {
      const int b = i / k;
      const double cos_t = trig[2 * b], sin_t = trig[2 * b + 1];
      const double rad = range_res_half + range_res * range;
      xyi[3 * o + 0] = (float)(rad * cos_t);  // :329
      xyi[3 * o + 1] = (float)(rad * sin_t);  // :330
      xyi[3 * o + 2] = (float)CFEAR_SLOT_INTENSITY(s);
      o++;
    }
  }
  __syncthreads();
  return total < cap ? total : cap;
}

// Compensate (utils.cpp:96-107) with GetRelTimeStamp (utils.h:28-32)
__device__ inline void compensate_block(float* __restrict__ xyi, int n, double m0, double m1, double m2, int ccw) {
  for (int i = threadIdx.x; i < n; i += CFEAR_FEAT_BLOCK) {
    const double px = (double)xyi[3 * i], py = (double)xyi[3 * i + 1];
    const double a = atan2(py, px);
    const double dd = ((a > 0.00001 ? a : (CFEAR_TWO_PI + a)) / CFEAR_TWO_PI);
    const double d = ccw ? -(dd - 0.5) : (dd - 0.5);
    const double s1 = sin(d * m2), c1 = cos(d * m2);
    const double tx = d * m0, ty = d * m1;
    xyi[3 * i + 0] = (float)((c1 * px + (-s1) * py) + tx);
    xyi[3 * i + 1] = (float)((s1 * px + c1 * py) + ty);
  }
  __syncthreads();
}

// cloud_build_block + compensate_block + the bounding box the voxel grid needs, in one pass over the slots
// (batched odometry step). bounds = {min x, max x, min y, max y} of the final points.
//
// Compensation needs atan2(y, x), sin and cos per point in f64 (utils.cpp:96-107) - hundreds of instructions each.
// Here the points of bearing b lie on the ray theta_b up to the float rounding of (x, y), so these are expanded
// around per-bearing values kept in LDS (tab: 4 doubles per bearing: angle, sin, cos of d_b * m2, d_b):
//   atan2(y, x) = theta_b + atan((y c_b - x s_b) / (x c_b + y s_b)),  argument ~1e-7 => atan(t) = t to f64 precision
//   sin/cos(arg) around arg_b = d_b * m2 to first order in (arg - arg_b) ~ 1e-9 (the second-order term is 1e-18)
// which agrees with evaluating the reference's expressions to within f64 rounding (the test tolerance on
// compensated points stays one float ulp, as for any two libm implementations). The sweep fraction a / (2 pi) is a
// multiplication by the rounded reciprocal here (a double division costs some 35 instructions per point): one ulp of d,
// 1e-16 of the point's coordinates before they are rounded to float.
// Compensate (utils.cpp:96-107) of one point as written: atan2, sweep fraction, rotation by the scaled motion
__device__ __noinline__ float2 compensate_point_plain(float x, float y, double m0, double m1, double m2, int ccw) {
  const double px = (double)x, py = (double)y;
  const double a = atan2(py, px);
  const double dd = ((a > 0.00001 ? a : (CFEAR_TWO_PI + a)) / CFEAR_TWO_PI);
  const double d = ccw ? -(dd - 0.5) : (dd - 0.5);
  double s1, c1;
  sincos(d * m2, &s1, &c1);
  return make_float2((float)((c1 * px + (-s1) * py) + d * m0), (float)((s1 * px + c1 * py) + d * m1));
}

// Compensate of a point on bearing b through the per-bearing table row tb = {angle, sin, cos of d_b * m2, d_b, ..} (the
// expansion described above); cb / sb: cos / sin of the bearing, rad: the point's range
template <typename TabPtr>
__device__ __forceinline__ void compensate_point_tabbed(float& x, float& y, TabPtr tb, double cb, double sb, double rad,
                                                        double m0, double m1, double m2, int ccw) {
  const double px = (double)x, py = (double)y;
  const double ab = tb[0], s_b = tb[1], c_b = tb[2], d_b = tb[3];
  const double a = ab + (py * cb - px * sb) * __builtin_amdgcn_rcp(rad);
  const double dd = (a > 0.00001 ? a : (CFEAR_TWO_PI + a)) * CFEAR_INV_TWO_PI;
  const double d = ccw ? -(dd - 0.5) : (dd - 0.5);
  const double e = (d - d_b) * m2;  // ~1e-9 * m2: first order in e is exact to double precision
  const double s1 = s_b + e * c_b, c1 = c_b - e * s_b;
  x = (float)((c1 * px + (-s1) * py) + d * m0);
  y = (float)((s1 * px + c1 * py) + d * m1);
}

// The points of a block in registers, handed from the cloud pass to the feature build. Wave w holds a contiguous run of the
// cloud, round after round in index order: the point of lane l in round r has index wbase + (points of the wave's earlier
// rounds) + (points of lower lanes in round r) - the order the stable counting sort of the feature build relies on.
#define CFEAR_PT 10  // rounds of 64 points a wave holds at most (A * k <= 5120 with 8 waves: every reference configuration)
struct PointRegs {
  float x[CFEAR_PT], y[CFEAR_PT];
  int wi[CFEAR_PT];                    // intensity (bits 0..7; the compact feature path takes integer intensities 0..255 only)
                                       // | index of the point in the cloud << 8
  unsigned long long bal[CFEAR_PT];    // wave-uniform: the lanes that hold a point in round r
  unsigned onm;                        // bit r: this lane holds a point in round r
  int wbase;                           // index of the wave's first point
  int rounds;                          // rounds in use (block-uniform); 0: the cloud does not fit, the points are not here
};
__device__ __forceinline__ bool preg_on(const PointRegs& R, int r) { return ((R.onm >> r) & 1u) != 0; }
__device__ __forceinline__ int preg_idx(const PointRegs& R, int r) { return R.wi[r] >> 8; }
__device__ __forceinline__ int preg_w(const PointRegs& R, int r) { return R.wi[r] & 255; }
__device__ __forceinline__ int preg_idx_from_ballots(const PointRegs& R, int r) {  // r: a compile-time constant in an unrolled loop
  int run = R.wbase;
#pragma unroll
  for (int q = 0; q < CFEAR_PT; q++) run += q < r ? __popcll(R.bal[q]) : 0;
  return run + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(R.bal[r] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)R.bal[r], 0u));
}
// a cloud in memory -> registers (per-call feature builds; the general cloud pass)
__device__ __forceinline__ void point_regs_from_global(const float* __restrict__ xyi, int n, PointRegs& R) {
  const __attribute__((address_space(1))) float* const g = (const __attribute__((address_space(1))) float*)xyi;
  const int wv = threadIdx.x >> 6, ln = lane_id(), nwv = CFEAR_FEAT_BLOCK >> 6;
  const int rounds = (n + 64 * nwv - 1) / (64 * nwv);
  R.rounds = rounds <= CFEAR_PT ? rounds : 0;
  R.wbase = wv * rounds * 64; R.onm = 0u;
#pragma unroll
  for (int r = 0; r < CFEAR_PT; r++) {
    const int t = (wv * rounds + r) * 64 + ln;
    const bool on = (r < R.rounds) & (t < n);
    const int tt = on ? t : 0;
    R.bal[r] = __ballot(on);
    R.onm |= on ? 1u << r : 0u;
    R.x[r] = g[3 * tt]; R.y[r] = g[3 * tt + 1]; R.wi[r] = ((int)g[3 * tt + 2] & 255) | (tt << 8);
  }
}

// registers -> cloud in memory (for a consumer that reads S->xyi after a cloud pass that kept it in registers only); block-collective
__device__ __forceinline__ void point_regs_to_global(const PointRegs& R, float* __restrict__ xyi) {
  __attribute__((address_space(1))) float* const g = (__attribute__((address_space(1))) float*)xyi;
  // (scalar stores: built as a vector, the optimizer widens the reads of R.x[r] / R.y[r] into loads that span the next element,
  // which moves the whole register array to scratch memory - for every user of PointRegs in the kernel)
#pragma unroll
  for (int r = 0; r < CFEAR_PT; r++) {
    const int o = 3 * preg_idx(R, r);
    const float x = R.x[r], y = R.y[r], w = (float)preg_w(R, r);
    if (preg_on(R, r)) { g[o] = x; g[o + 1] = y; g[o + 2] = w; }
  }
  __syncthreads();
}

// slots of one sweep -> compensated cloud (getPeaksFilteredPointCloud, radar_filters.cpp:309-337, + Compensate, utils.cpp:96-107)
// in S->xyi AND in the registers of the block (PR), with the bounding box. Returns the number of points.
__device__ inline int cloud_step_block(const uint32_t* __restrict__ slots, int A, int k, const double* __restrict__ trig,
                                       float range_res_f, float min_distance_f, float* __restrict__ xyi, int cap, int compensate,
                                       double m0, double m1, double m2, int ccw, int* red_i, float* red_f, double* tab,
                                       int tab_bearings, float bounds[4], PointRegs& PR, bool store = true) {
  // store = false: the caller can do without the cloud in S->xyi when the points are handed over in registers - 58 KB per
  // sweep that the compact feature path never reads back (the general branch below always writes it);
  // point_regs_to_global writes it later for a caller that turns out to need it.
  // the sweep's slots, the trigonometric table and the cloud are global arrays: global-typed pointers give global_load /
  // global_store instead of flat instructions (which also count against the LDS counter)
  const __attribute__((address_space(1))) uint32_t* const g_slots = (const __attribute__((address_space(1))) uint32_t*)slots;
  const __attribute__((address_space(1))) double* const g_trig = (const __attribute__((address_space(1))) double*)trig;
  __attribute__((address_space(1))) float* const g_xyi = (__attribute__((address_space(1))) float*)xyi;
  const double range_res = (double)range_res_f;
  const int min_range_bin = (int)ceil((double)min_distance_f / range_res);  // radar_filters.cpp:315
  const double range_res_half = range_res / 2.0;
  const int items = A * k;
  const int wv = threadIdx.x >> 6, ln = lane_id(), nwv = CFEAR_FEAT_BLOCK >> 6;
  // per-bearing table in LDS, six doubles each: principal angle, sin / cos of the compensation rotation at the bearing's
  // angle, its sweep fraction, cos / sin of the bearing (so that the per-point loop has no global load to wait for)
  auto* ltab = CFEAR_LDS_PTR(double, tab);
  const bool tabbed = A <= tab_bearings;       // block-uniform; more bearings than the table holds: plain formulas
  auto build_table = [&]() {
   if (tabbed) {
    for (int b = threadIdx.x; b < A; b += CFEAR_FEAT_BLOCK) {
      ltab[6 * b + 4] = g_trig[2 * b]; ltab[6 * b + 5] = g_trig[2 * b + 1];
      if (compensate) {
        const double theta = ((double)(b + 1) / A) * CFEAR_TWO_PI;                   // radar_filters.cpp:317
        const double a = theta > 3.14159265358979323846 ? theta - CFEAR_TWO_PI : theta;  // principal value, as atan2 returns it
        const double dd = ((a > 0.00001 ? a : (CFEAR_TWO_PI + a)) / CFEAR_TWO_PI);   // utils.h:28-32
        const double d = ccw ? -(dd - 0.5) : (dd - 0.5);
        double sb, cb;
        sincos(d * m2, &sb, &cb);
        ltab[6 * b] = a; ltab[6 * b + 1] = sb; ltab[6 * b + 2] = cb; ltab[6 * b + 3] = d;
      }
    }
  }
  };
  float mnx = 3.4e38f, mxx = -3.4e38f, mny = 3.4e38f, mxy = -3.4e38f;
  const int RC = (items + 64 * nwv - 1) / (64 * nwv);  // rounds of 64 slots per wave
  int total;
  if (tabbed && RC <= CFEAR_PT && k < 65536 && items < 65536) {  // block-uniform: every reference configuration
    // Wave w takes the slots [w * RC * 64, (w + 1) * RC * 64) round after round, lane <-> slot: coalesced loads, and the
    // kept points of the wave are contiguous in the cloud, in register order (see PointRegs).
    // t / k = umulhi(t, magic) for t < 65536 and 2 <= k < 65536 (checked exhaustively for the supported k = 2..64); for k = 1 the
    // magic number would be 2^32, which does not fit: the quotient is t itself
    const unsigned magic = k > 1 ? (unsigned)((0x100000000ull + (unsigned)k - 1u) / (unsigned)k) : 0u;
    uint32_t sv[CFEAR_PT];
#pragma unroll
    for (int r = 0; r < CFEAR_PT; r++) {  // the slots are on their way from memory while the table is built
      const int t = (wv * RC + r) * 64 + ln;
      sv[r] = g_slots[((r < RC) & (t < items)) ? t : 0];
    }
    build_table();
    int wtot = 0;
    PR.onm = 0u;
#pragma unroll
    for (int r = 0; r < CFEAR_PT; r++) {
      const int t = (wv * RC + r) * 64 + ln;
      const bool inr = (r < RC) & (t < items);
      const bool on = inr & (CFEAR_SLOT_VALID(sv[r]) != 0) & (CFEAR_SLOT_RANGE(sv[r]) > min_range_bin);  // :327
      PR.bal[r] = __ballot(on);
      PR.onm |= on ? 1u << r : 0u;
      wtot += __popcll(PR.bal[r]);
    }
    {  // index of the wave's first point (this barrier also publishes the table; the first use of the scan scratch)
      auto* sl = CFEAR_LDS_PTR(int, red_i);
      if (ln == 0) sl[wv] = wtot;
      __syncthreads();
      int base = 0; total = 0;
      for (int i = 0; i < nwv; i++) { const int c = sl[i]; base += i < wv ? c : 0; total += c; }
      PR.wbase = base;
    }
    if (total > cap) {  // block-uniform, never with the capacities the library allocates: keep the first cap points
      bool keep[CFEAR_PT];
#pragma unroll
      for (int r = 0; r < CFEAR_PT; r++) keep[r] = preg_on(PR, r) && preg_idx_from_ballots(PR, r) < cap;
      PR.onm = 0u;
#pragma unroll
      for (int r = 0; r < CFEAR_PT; r++) { PR.bal[r] = __ballot(keep[r]); PR.onm |= keep[r] ? 1u << r : 0u; }
    }
    PR.rounds = RC;
#pragma unroll
    for (int r = 0; r < CFEAR_PT; r++) {  // branch-free per round (independent chains of double-precision operations interleave)
      const uint32_t s = sv[r];
      const bool on = preg_on(PR, r);
      const int range = CFEAR_SLOT_RANGE(s);
      const int t = (wv * RC + r) * 64 + ln;
      const int bb = min(k > 1 ? (int)__umulhi((unsigned)t, magic) : t, A - 1);
      const double cb = ltab[6 * bb + 4], sb = ltab[6 * bb + 5];
      const double rad = range_res_half + range_res * range;
      float x = (float)(rad * cb);  // :329
      float y = (float)(rad * sb);  // :330
      if (compensate) {  // block-uniform
        // utils.cpp:96-107 expanded around the bearing (see above). The angle of (x, y) off the bearing's ray - the float
        // rounding of x and y, ~1e-8 rad - is (y c_b - x s_b) / (x c_b + y s_b); its denominator is the range to 1e-8 and two
        // digits of the quotient are plenty (it moves the point by ~1e-9 m): a hardware reciprocal of the range
        compensate_point_tabbed(x, y, ltab + 6 * bb, cb, sb, rad, m0, m1, m2, ccw);
      }
      const int o = preg_idx_from_ballots(PR, r);
      PR.x[r] = x; PR.y[r] = y; PR.wi[r] = (int)CFEAR_SLOT_INTENSITY(s) | (o << 8);
      if (on) {
        if (store) {  // one 12-byte store (block-uniform condition)
          typedef float f32x3 __attribute__((ext_vector_type(3)));
          typedef f32x3 __attribute__((aligned(4))) f32x3u;
          *(__attribute__((address_space(1))) f32x3u*)(g_xyi + 3 * o) = f32x3{x, y, (float)CFEAR_SLOT_INTENSITY(s)};
        }
        mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
      }
    }
    total = total < cap ? total : cap;
    bounds[0] = mnx; bounds[1] = mxx; bounds[2] = mny; bounds[3] = mxy;
    block_bounds<CFEAR_FEAT_BLOCK>(bounds, red_f);  // (its barriers come after every store of the cloud: the general feature path may read it back)
    return total;
  }
  // general: any number of slots. Wave w takes the slots [w * RC * 64, (w + 1) * RC * 64) as above, lane <-> slot, but in a
  // loop of batches of eight rounds (eight independent coalesced loads in flight per lane; a thread walking a run of slots of
  // its own waits for one load after the other): a counting pass, the scan over the waves, then the points; the cloud goes to
  // memory and comes back
  build_table();
  const bool small = k > 1 && k < 65536 && items < 65536;  // block-uniform: t / k by multiplication (see above)
  const unsigned magic = small ? (unsigned)((0x100000000ull + (unsigned)k - 1u) / (unsigned)k) : 0u;
  const int t_wave = wv * RC * 64;
  int wtot = 0;
  for (int r0 = 0; r0 < RC; r0 += 8) {
    uint32_t sv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int t = t_wave + (r0 + u) * 64 + ln;
      sv[u] = g_slots[((r0 + u < RC) & (t < items)) ? t : 0];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int t = t_wave + (r0 + u) * 64 + ln;
      const bool on = (r0 + u < RC) & (t < items) & (CFEAR_SLOT_VALID(sv[u]) != 0) & (CFEAR_SLOT_RANGE(sv[u]) > min_range_bin);  // :327
      wtot += __popcll(__ballot(on));
    }
  }
  int run;
  {  // index of the wave's first point (this barrier also publishes the table)
    auto* sl = CFEAR_LDS_PTR(int, red_i);
    if (ln == 0) sl[wv] = wtot;
    __syncthreads();
    int base = 0; total = 0;
    for (int i = 0; i < nwv; i++) { const int c = sl[i]; base += i < wv ? c : 0; total += c; }
    run = base;
  }
  // (the block-uniform choices - bearing table, slot -> bearing by multiplication - as compile-time constants of the loop body: tested
  // inside it they doubled its instruction count; the usual case is both)
  auto points_pass = [&](auto tab_c, auto small_c) {
    constexpr bool TAB = decltype(tab_c)::value, SMALL = decltype(small_c)::value;
    for (int r0 = 0; r0 < RC; r0 += 8) {
      uint32_t sv[8];
  #pragma unroll
      for (int u = 0; u < 8; u++) {
        const int t = t_wave + (r0 + u) * 64 + ln;
        sv[u] = g_slots[((r0 + u < RC) & (t < items)) ? t : 0];
      }
  #pragma unroll
      for (int u = 0; u < 8; u++) {
        const int t = t_wave + (r0 + u) * 64 + ln;
        const uint32_t s = sv[u];
        const int range = CFEAR_SLOT_RANGE(s);
        const bool hit = (r0 + u < RC) & (t < items) & (CFEAR_SLOT_VALID(s) != 0) & (range > min_range_bin);
        const unsigned long long bal = __ballot(hit);
        const int o = run + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        run += __popcll(bal);
        if (hit && o < cap) {
          const int b = min(SMALL ? (int)__umulhi((unsigned)t, magic) : (int)((unsigned)t / (unsigned)k), A - 1);
          const double cb = TAB ? ltab[6 * b + 4] : g_trig[2 * b], sb = TAB ? ltab[6 * b + 5] : g_trig[2 * b + 1];
          const double rad = range_res_half + range_res * range;
          float x = (float)(rad * cb);  // :329
          float y = (float)(rad * sb);  // :330
          if (compensate) {
            if (TAB) compensate_point_tabbed(x, y, ltab + 6 * b, cb, sb, rad, m0, m1, m2, ccw);
            else {  // utils.cpp:96-107 as written (out of line)
              const float2 p = compensate_point_plain(x, y, m0, m1, m2, ccw);
              x = p.x; y = p.y;
            }
          }
          typedef float f32x3 __attribute__((ext_vector_type(3)));
          typedef f32x3 __attribute__((aligned(4))) f32x3u;
          *(__attribute__((address_space(1))) f32x3u*)(g_xyi + 3 * o) = f32x3{x, y, (float)CFEAR_SLOT_INTENSITY(s)};
          mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
        }
      }
    }
  };
  if (tabbed && small) points_pass(std::true_type{}, std::true_type{});
  else if (tabbed) points_pass(std::true_type{}, std::false_type{});
  else if (small) points_pass(std::false_type{}, std::true_type{});
  else points_pass(std::false_type{}, std::false_type{});
  total = total < cap ? total : cap;
  bounds[0] = mnx; bounds[1] = mxx; bounds[2] = mny; bounds[3] = mxy;
  block_bounds<CFEAR_FEAT_BLOCK>(bounds, red_f);
  __syncthreads();
  point_regs_from_global(xyi, total, PR);
  return total;
}

// closed-form symmetric 2x2 eigen-decomposition; identical formulas to the oracle's eig2()
__device__ inline void eig2(double a, double b, double c, double* lmin, double* lmax, double vmin[2], double vmax[2]) {
  const double t1 = 0.5 * (a + c);
  const double d = 0.5 * (a - c);
  const double t0 = sqrt(d * d + b * b);
  *lmin = t1 - t0;
  *lmax = t1 + t0;
  const double v0x = *lmax - c, v0y = b;
  const double v1x = b, v1y = *lmax - a;
  const double n0 = v0x * v0x + v0y * v0y, n1 = v1x * v1x + v1y * v1y;
  double vx, vy, nn;
  if (n0 >= n1) { vx = v0x; vy = v0y; nn = n0; } else { vx = v1x; vy = v1y; nn = n1; }
  if (!(nn > 0.0)) { vmax[0] = 0; vmax[1] = 1; vmin[0] = 1; vmin[1] = 0; return; }
  const double inv = 1.0 / sqrt(nn);
  vmax[0] = vx * inv; vmax[1] = vy * inv;
  vmin[0] = -vmax[1]; vmin[1] = vmax[0];
}

template <typename IntPtr>
__device__ inline int lower_bound_int(IntPtr a, int n, long long v) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if ((long long)a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}

// shifted weighted moments of the points q in [a, b) of the sorted array that lie within r2 of (cx, cy)
struct CellAcc { int m; double s0, s1x, s1y, sxx, sxy, syy; };
// lane `sub` of a group of GS lanes takes the batches a + 4*sub, a + 4*(sub + GS), ... (the chunked caller uses sub = 0, GS = 1)
template <typename FloatPtr>
__device__ __forceinline__ void accumulate_range(FloatPtr sp, int a, int b, float cx, float cy, float r2,
                                        int weight_intensity, CellAcc& A, int sub, int GS) {
  const double cxd = (double)cx, cyd = (double)cy;
  for (int q = a + 4 * sub; q < b; q += 4 * GS) {
    float px[4], py[4], pw[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {  // independent LDS loads first, arithmetic afterwards
      const int qq = min(q + u, b - 1);
      px[u] = sp[3 * qq]; py[u] = sp[3 * qq + 1]; pw[u] = sp[3 * qq + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const float dx = cx - px[u], dy = cy - py[u];
      float d2 = dx * dx; d2 += dy * dy;
      if (q + u < b && d2 < r2) {  // pointnormal.cpp:291 radius test (float, strict)
        const double w = weight_intensity ? fmax((double)pw[u] - 60.0, 0.0) : 1.0;  // :15
        const double ex = (double)px[u] - cxd, ey = (double)py[u] - cyd;
        A.m++; A.s0 += w; A.s1x += w * ex; A.s1y += w * ey;
        A.sxx += w * (ex * ex); A.sxy += w * (ex * ey); A.syy += w * (ey * ey);
      }
    }
  }
}

// ComputeSearchTreeFromCells (pointnormal.cpp:151-162): uniform grid over the float cell means of the nc cells already in
// S (cells / mean_f / rsrc / rtar written by the caller). Block-collective; lm_ok: the float means also sit in the LDS copy
// the feature epilogue left in W.vlist.
__device__ __forceinline__ void cell_grid_block(ScanDev* __restrict__ S, int nc, const FeatureParams& P, const FeatureScratch& W,
                                                bool lm_ok, PhaseTimer* pt) {
  typedef __attribute__((address_space(1))) float g_f32;
  typedef __attribute__((address_space(1))) int g_i32;
  const int tid = threadIdx.x, nt = CFEAR_FEAT_BLOCK;
  // ---- uniform grid over the float cell means (replaces KdTreeFLANN<PointXY>, :151-162) ----
  // the cell means come from the LDS copy the epilogue left in the voxel-list array when they fit (no read-back from memory)
  const auto* lmr = CFEAR_LDS_PTR(float, reinterpret_cast<float*>(W.vlist));
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  g_f32* const g_mean = (g_f32*)S->mean_f;
  g_i32* const g_gstart = (g_i32*)S->gstart;
  __attribute__((address_space(1))) f32x4* const g_gpts = (__attribute__((address_space(1))) f32x4*)S->gpts;
  float gx0 = 3.4e38f, gx1 = -3.4e38f, gy0 = 3.4e38f, gy1 = -3.4e38f;
  for (int i = tid; i < nc; i += nt) {
    const float x = lm_ok ? lmr[2 * i] : g_mean[2 * i], y = lm_ok ? lmr[2 * i + 1] : g_mean[2 * i + 1];
    gx0 = fminf(gx0, x); gx1 = fmaxf(gx1, x); gy0 = fminf(gy0, y); gy1 = fmaxf(gy1, y);
  }
  { float bb[4] = {gx0, gx1, gy0, gy1}; block_bounds<CFEAR_FEAT_BLOCK>(bb, W.red_f); gx0 = bb[0]; gx1 = bb[1]; gy0 = bb[2]; gy1 = bb[3]; }
  if (nc == 0) {
    if (tid == 0) { S->gw = 0; S->gh = 0; S->gcell = 1.f; S->gminx = 0.f; S->gminy = 0.f; }
    __syncthreads();
    return;
  }
  float gcell = (float)(2.0 * P.assoc_radius);
  int gw, gh;
  for (int it = 0;; it++) {
    gw = (int)floorf((gx1 - gx0) / gcell) + 1;
    gh = (int)floorf((gy1 - gy0) / gcell) + 1;
    if (gw >= 1 && gh >= 1 && (long long)gw * gh <= S->cap_grid) break;
    gcell *= 2.f;
    if (it >= 64 || !(gcell > 0.f) || !isfinite(gcell)) {  // non-finite extents (or a non-positive cell size): one bucket, never a spin
      gw = 1; gh = 1; gcell = 3.0e38f;
      break;
    }
  }
  const int G = gw * gh;
  typedef __attribute__((address_space(1))) unsigned short g_u16;
  g_u16* const g_off16 = (g_u16*)grid_off16(S->gstart);
  const bool small = nc <= CFEAR_GRID16_MAX;  // block-uniform: 16-bit offsets (see grid_off16)
  if (W.lds && G + 1 <= W.tab_voxels / 2 && small) {
    // bucket counters / cursors in LDS (the key region: the staged points are not needed any more)
    int* gc = reinterpret_cast<int*>(W.keys);
    for (int g = tid; g <= G; g += nt) gc[g] = 0;
    __syncthreads();
    for (int i = tid; i < nc; i += nt) {
      const float mx = lm_ok ? lmr[2 * i] : g_mean[2 * i], my = lm_ok ? lmr[2 * i + 1] : g_mean[2 * i + 1];
      int cx = (int)floorf((mx - gx0) / gcell), cy = (int)floorf((my - gy0) / gcell);
      cx = min(max(cx, 0), gw - 1); cy = min(max(cy, 0), gh - 1);
      atomicAdd(&gc[cy * gw + cx + 1], 1);
    }
    __syncthreads();
    {
      const int ipt = (G + nt - 1) / nt;
      const int i0 = tid * ipt, i1 = min(G, i0 + ipt);
      int cnt = 0;
      for (int g = i0; g < i1; g++) cnt += gc[g + 1];
      int tot;
      int o = block_exclusive_scan<CFEAR_FEAT_BLOCK>(cnt, W.red_i, &tot);
      for (int g = i0; g < i1; g++) {  // cursor in LDS; the offset of the next bucket goes out as 16 bits
        const int c = gc[g + 1]; gc[g + 1] = o; o += c; g_off16[g + 1] = (unsigned short)o;
      }
      if (tid == 0) g_off16[0] = 0;
      for (int g = G + 1 + tid; g <= G + 2 * gw; g += nt) g_off16[g] = (unsigned short)nc;  // the padding behind the last bucket
      __syncthreads();
    }
    for (int i = tid; i < nc; i += nt) {
      const float mx = lm_ok ? lmr[2 * i] : g_mean[2 * i], my = lm_ok ? lmr[2 * i + 1] : g_mean[2 * i + 1];
      int cx = (int)floorf((mx - gx0) / gcell), cy = (int)floorf((my - gy0) / gcell);
      cx = min(max(cx, 0), gw - 1); cy = min(max(cy, 0), gh - 1);
      const int pos = atomicAdd(&gc[cy * gw + cx + 1], 1);
      g_gpts[pos] = f32x4{mx, my, __int_as_float(i), 0.f};
    }
  } else {
  for (int g = tid; g <= G; g += nt) g_gstart[g] = 0;
  __syncthreads();
  for (int i = tid; i < nc; i += nt) {
    int cx = (int)floorf((g_mean[2 * i] - gx0) / gcell), cy = (int)floorf((g_mean[2 * i + 1] - gy0) / gcell);
    cx = min(max(cx, 0), gw - 1); cy = min(max(cy, 0), gh - 1);
    atomicAdd(&S->gstart[cy * gw + cx + 1], 1);
  }
  __syncthreads();
  {  // exclusive scan of the bucket counts (gstart[g+1] holds count of bucket g)
    const int ipt = (G + nt - 1) / nt;
    const int i0 = tid * ipt, i1 = min(G, i0 + ipt);
    int cnt = 0;
    for (int g = i0; g < i1; g++) cnt += g_gstart[g + 1];
    int tot;
    int o = block_exclusive_scan<CFEAR_FEAT_BLOCK>(cnt, W.red_i, &tot);
    for (int g = i0; g < i1; g++) { const int c = g_gstart[g + 1]; W.vcur[g] = o; g_gstart[g + 1] = o + c; o += c; }
    __syncthreads();
  }
  // scatter cell indices into their buckets (order inside a bucket is irrelevant: the query breaks
  // exact ties by cell index)
  for (int i = tid; i < nc; i += nt) {
    int cx = (int)floorf((g_mean[2 * i] - gx0) / gcell), cy = (int)floorf((g_mean[2 * i + 1] - gy0) / gcell);
    cx = min(max(cx, 0), gw - 1); cy = min(max(cy, 0), gh - 1);
    const int pos = atomicAdd(&W.vcur[cy * gw + cx], 1);
    g_gpts[pos] = f32x4{g_mean[2 * i], g_mean[2 * i + 1], __int_as_float(i), 0.f};
  }
  if (small) {  // the 16-bit copy every reader of a small scan uses (the offsets in gstart are final since the barrier before the scatter)
    for (int g = tid; g <= G + 2 * gw; g += nt) g_off16[g] = (unsigned short)g_gstart[min(g, G)];
  }
  }
  if (tid == 0) { S->gminx = gx0; S->gminy = gy0; S->gcell = gcell; S->gw = gw; S->gh = gh; }
  __syncthreads();
  if (pt) pt->mark();
}

// MapPointNormal::ComputeNormals + ComputeSearchTreeFromCells for the cloud already in S->xyi: the general path (any
// cloud size, any voxel grid, any intensities) in global arrays. p2 = power of two >= n with p2 <= capacity of W.keys.
__device__ __forceinline__ void features_block(ScanDev* __restrict__ S, int n, const FeatureParams& P, const FeatureScratch& W, int p2,
                                      PhaseTimer* pt = nullptr, const float* bounds = nullptr) {
  // global working arrays through global-typed pointers: global_load / global_store instead of flat instructions (a flat
  // access also counts against the LDS counter, so waiting for LDS data would wait for it as well)
  typedef __attribute__((address_space(1))) double g_f64;
  typedef __attribute__((address_space(1))) float g_f32;
  typedef __attribute__((address_space(1))) int g_i32;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) i32x4 g_i32x4;
  g_f64* const g_part = (g_f64*)W.part;
  g_f32* const g_samples = (g_f32*)W.samples;
  g_i32* const g_rng = (g_i32*)W.rng;
  typedef __attribute__((address_space(1))) uint64_t g_u64;
  g_u64* const g_keys = (g_u64*)W.keys;      // (every caller hands this path arrays in memory: make_fscratch)
  g_i32* const g_order = (g_i32*)W.order;
  g_i32* const g_vstart = (g_i32*)W.vstart;
  g_i32* const g_vlist = (g_i32*)W.vlist;
  g_i32* const g_tmpi = (g_i32*)W.tmpi;
  g_f32* const g_spts = (g_f32*)W.spts;
  const int tid = threadIdx.x, nt = CFEAR_FEAT_BLOCK;
  const g_f32* const xyi = (const g_f32*)S->xyi;
  if (n <= 0) {  // reference: exit(0) (pointnormal.cpp:72-75)
    if (tid == 0) { S->n_points = 0; S->n_samples = 0; S->n_cells = 0; S->status = CFEAR_ERR_EMPTY; S->gw = 0; S->gh = 0; }
    __syncthreads();
    return;
  }
  // ---- PCL VoxelGrid (pointnormal.cpp:277-280), leaf = radius_/downsample_factor ----
  const float leaf = (float)((double)P.radius / P.downsample_factor);
  const float inv = 1.0f / leaf;
  float mnx = 3.4e38f, mxx = -3.4e38f, mny = 3.4e38f, mxy = -3.4e38f;
  if (bounds) {  // block-uniform: the caller already knows the bounding box
    mnx = bounds[0]; mxx = bounds[1]; mny = bounds[2]; mxy = bounds[3];
  } else {
    for (int i = tid; i < n; i += nt) {
      const float x = xyi[3 * i], y = xyi[3 * i + 1];
      mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
    }
    float bb[4] = {mnx, mxx, mny, mxy};
    block_bounds<CFEAR_FEAT_BLOCK>(bb, W.red_f);
    mnx = bb[0]; mxx = bb[1]; mny = bb[2]; mxy = bb[3];
  }
  const int min_b0 = (int)floorf(mnx * inv), max_b0 = (int)floorf(mxx * inv);
  const int min_b1 = (int)floorf(mny * inv), max_b1 = (int)floorf(mxy * inv);
  const int div0 = max_b0 - min_b0 + 1, div1 = max_b1 - min_b1 + 1;
  int nv_out = 0;
  {
  for (int i = tid; i < p2; i += nt) {
    uint64_t key = ~0ull;
    if (i < n) {
      const int ijk0 = (int)(floorf(xyi[3 * i] * inv) - (float)min_b0);
      const int ijk1 = (int)(floorf(xyi[3 * i + 1] * inv) - (float)min_b1);
      const uint64_t idx = (uint64_t)((long long)ijk0 + (long long)ijk1 * (long long)div0);
      key = (idx << 32) | (uint64_t)(uint32_t)(W.vrank ? W.vrank[i] : i);  // (inside a voxel: by point index, or by the given order)
    }
    g_keys[i] = key;
  }
  if (pt) pt->mark();
  block_bitonic_sort(g_keys, p2);  // [3P] std::sort on the voxel index, pinned as stable by the point index
  if (pt) pt->mark();
  // ---- voxel segments, sorted order ----
  {
    const int ipt = (n + nt - 1) / nt;
    const int i0 = tid * ipt, i1 = min(n, i0 + ipt);
    int cnt = 0;
    for (int i = i0; i < i1; i++) cnt += (i == 0 || (g_keys[i] >> 32) != (g_keys[i - 1] >> 32)) ? 1 : 0;
    int nv;
    int o = block_exclusive_scan<CFEAR_FEAT_BLOCK>(cnt, W.red_i, &nv);
    for (int i = i0; i < i1; i++) {
      const uint64_t k = g_keys[i];
      g_order[i] = W.vperm ? W.vperm[(uint32_t)k] : (int)(uint32_t)k;
      if (i == 0 || (k >> 32) != (g_keys[i - 1] >> 32)) { g_vstart[o] = i; g_vlist[o] = (int)(k >> 32); o++; }
    }
    nv_out = nv;
    if (tid == 0) { g_vstart[nv] = n; S->n_samples = nv; S->n_points = n; S->status = 0; }
    __syncthreads();
  }
  }
  const int nv = nv_out;  // known to every thread: no read-back of S->n_samples through memory
  // stage the points in sorted order (over the key region when it is in LDS: every key has been consumed)
  for (int q = tid; q < n; q += nt) {
    const int pi = g_order[q];
    g_spts[3 * q] = xyi[3 * pi]; g_spts[3 * q + 1] = xyi[3 * pi + 1]; g_spts[3 * q + 2] = xyi[3 * pi + 2];
  }
  __syncthreads();
  if (pt) pt->mark();
  const g_f32* const sp = g_spts;
  // ---- centroids: float sums in ascending (voxel, point) order, divided by float(count) ([3P] PCL CentroidPoint) ----
  // (the same loop goes on to the sample's candidate ranges below: the centroid stays in registers instead of being read
  // back from memory behind a barrier)
  if (pt) pt->mark();
  // ---- radius search + cell statistics per sample point (pointnormal.cpp:286-296, :7-63) ----
  // One pass over the candidates with moments shifted by the sample point c:
  //   mean = c + S1/S0,  cov = S2/S0 - (S1/S0)(S1/S0)^T   (== sum w_i (x_i-u)(x_i-u)^T with sum w_i = 1)
  const float r2 = (float)((double)P.radius * (double)P.radius);
  const float rq = P.radius * 1.0001f;
  // Candidate counts range from 1 to ~1000 per sample point, so the work is cut into chunks of at most C
  // candidates: (1) per sample the candidate row ranges and their total, (2) a scan turns them into a chunk
  // list, (3) one lane per chunk accumulates partial moments, (4) the epilogue adds a sample's partials in
  // chunk order (deterministic).
  g_i32* const T = g_order;  // the sorted order has been consumed by the staging above
  for (int v = tid; v < nv; v += nt) {
    float cx, cy;
    {
      const int a = g_vstart[v], b = g_vstart[v + 1];
      float sx = 0.f, sy = 0.f, si = 0.f;
      for (int q = a; q < b; q++) { sx += sp[3 * q]; sy += sp[3 * q + 1]; si += sp[3 * q + 2]; }
      const float cnt = (float)(b - a);
      cx = sx / cnt; cy = sy / cnt;
      g_samples[3 * v] = cx; g_samples[3 * v + 1] = cy; g_samples[3 * v + 2] = si / cnt;
    }
    int gx0 = (int)(floorf((cx - rq) * inv) - (float)min_b0), gx1 = (int)(floorf((cx + rq) * inv) - (float)min_b0);
    int gy0 = (int)(floorf((cy - rq) * inv) - (float)min_b1), gy1 = (int)(floorf((cy + rq) * inv) - (float)min_b1);
    gx0 = max(gx0, 0); gy0 = max(gy0, 0); gx1 = min(gx1, div0 - 1); gy1 = min(gy1, div1 - 1);
    int tot = 0, nr = 0;
    int R[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const long long vid = g_vlist[v];
    for (int gy = gy0; gy <= gy1 && gx0 <= gx1; gy++) {
      // voxels gx0..gx1 of this row are contiguous in the sorted order: search the voxel list
      const long long k0 = (long long)gx0 + (long long)gy * div0, k1 = (long long)gx1 + (long long)gy * div0;
      // sample v sits at position v of the voxel list (its own voxel): voxel indices are distinct integers, so the
      // first voxel >= k0 is at most |k0 - own index| positions away from v - a short search instead of all of the list
      int lo, hi;
      if (k0 <= vid) { lo = v - (int)min((long long)v, vid - k0); hi = v; }
      else { lo = v; hi = (int)min((long long)nv, (long long)v + (k0 - vid)); }
      const int p0 = lo + lower_bound_int(g_vlist + lo, hi - lo, k0);
      int p1 = p0;
      while (p1 < nv && (long long)g_vlist[p1] <= k1) p1++;
      const int a = g_vstart[p0], b = g_vstart[p1];
      if (b > a) {
#pragma unroll
        for (int u = 0; u < 4; u++)  // static indices: a run-time index would move R[] to per-thread scratch
          if (u == nr) { R[2 * u] = a; R[2 * u + 1] = b; }
        nr++; tot += b - a;
      }
    }
    if (nr > 4) R[0] = -1;  // only with leaf < radius: the chunk lanes search again (copies made below)
    {
      g_i32x4* Rg = (g_i32x4*)(g_rng + 8 * (size_t)v);
      Rg[0] = i32x4{R[0], R[1], R[2], R[3]}; Rg[1] = i32x4{R[4], R[5], R[6], R[7]};
    }
    T[v] = tot >= 6 ? tot : 0;  // fewer than six candidates can never make a cell (pointnormal.cpp:291): no chunks, no partial sums
  }
  const bool wide = (int)(2.0f * rq * inv) + 2 > 4;  // block-uniform
  if (wide) {
    for (int i = tid; i < nv; i += nt) { g_tmpi[i] = g_vlist[i]; g_tmpi[W.cap + 8 + i] = g_vstart[i]; }
    if (tid == 0) g_tmpi[W.cap + 8 + nv] = g_vstart[nv];
  }
  __syncthreads();
  if (pt) pt->mark();
  int C = 32, CS = 5, NC;  // (C = 1 << CS)
  {
    const int ipt = (nv + nt - 1) / nt;
    const int i0 = tid * ipt, i1 = min(nv, i0 + ipt);
    int o;
    for (;;) {  // block-uniform: double the chunk size until the chunk list fits
      int cnt = 0;
      for (int i = i0; i < i1; i++) cnt += (T[i] + C - 1) >> CS;
      o = block_exclusive_scan<CFEAR_FEAT_BLOCK>(cnt, W.red_i, &NC);
      if (NC <= W.cap) break;
      C <<= 1; CS++;
    }
    for (int i = i0; i < i1; i++) {  // vstart/vlist are free now: chunk start per sample, sample per chunk
      const int c = (T[i] + C - 1) >> CS;
      g_vstart[i] = o;
      for (int j = 0; j < c; j++) g_vlist[o + j] = i;
      o += c;
    }
    if (tid == 0) g_vstart[nv] = NC;
    __syncthreads();
  }
  if (pt) pt->mark();
  for (int w = tid; w < NC; w += nt) {
    const int v = g_vlist[w];
    const int j = w - g_vstart[v];
    float cx, cy;
    i32x4 r0, r1;
    {
      cx = g_samples[3 * v]; cy = g_samples[3 * v + 1];
      const g_i32x4* R = (const g_i32x4*)(g_rng + 8 * (size_t)v);
      r0 = R[0]; r1 = R[1];
    }
    const int tot = T[v];
    int skip = j * C, left = min(C, tot - skip);
    CellAcc A = {0, 0, 0, 0, 0, 0, 0};
    if (r0.x >= 0) {
      const int ra[4] = {r0.x, r0.z, r1.x, r1.z}, rb[4] = {r0.y, r0.w, r1.y, r1.w};
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int len = rb[r] - ra[r];
        if (left > 0 && len > 0) {
          if (skip >= len) { skip -= len; }
          else {
            const int s = ra[r] + skip, e = min(rb[r], s + left);
            accumulate_range(sp, s, e, cx, cy, r2, P.weight_intensity, A, 0, 1);
            left -= e - s; skip = 0;
          }
        }
      }
    } else {  // more than four candidate rows (leaf < radius): the ranges are searched again
      int gx0 = (int)(floorf((cx - rq) * inv) - (float)min_b0), gx1 = (int)(floorf((cx + rq) * inv) - (float)min_b0);
      int gy0 = (int)(floorf((cy - rq) * inv) - (float)min_b1), gy1 = (int)(floorf((cy + rq) * inv) - (float)min_b1);
      gx0 = max(gx0, 0); gy0 = max(gy0, 0); gx1 = min(gx1, div0 - 1); gy1 = min(gy1, div1 - 1);
      const g_i32* vl = g_tmpi;  // copies of vlist/vstart made before they were reused
      const g_i32* vs = g_tmpi + W.cap + 8;
      for (int gy = gy0; gy <= gy1 && left > 0; gy++) {
        const long long k0 = (long long)gx0 + (long long)gy * div0, k1 = (long long)gx1 + (long long)gy * div0;
        const int p0 = lower_bound_int(vl, nv, k0);
        int p1 = p0;
        while (p1 < nv && (long long)vl[p1] <= k1) p1++;
        const int a = vs[p0], b = vs[p1], len = b - a;
        if (skip >= len) { skip -= len; continue; }
        const int s = a + skip, e = min(b, s + left);
        accumulate_range(sp, s, e, cx, cy, r2, P.weight_intensity, A, 0, 1);
        left -= e - s; skip = 0;
      }
    }
    const size_t cs = (size_t)W.cap;
    g_part[w] = (double)A.m; g_part[cs + w] = A.s0; g_part[2 * cs + w] = A.s1x; g_part[3 * cs + w] = A.s1y;
    g_part[4 * cs + w] = A.sxx; g_part[5 * cs + w] = A.sxy; g_part[6 * cs + w] = A.syy;
  }
  __syncthreads();
  if (pt) pt->mark();
  // ---- cell epilogue + compaction: a cell is built in registers and, if it is valid, written straight to
  // its final slot; one block scan per round of blockDim samples keeps the sample order (pointnormal.cpp:292-294)
  int n_cells_out;  // every thread knows it (sum of the block scans): no read-back of S->n_cells through memory
  {
    int base = 0;
    const int cap_cells = S->cap_cells;
    const bool keep_cells = S->cells != nullptr;
    for (int v0 = 0; v0 < nv; v0 += nt) {
      const int v = v0 + tid;
      cfear_cell c;
      int valid = 0;
      if (v < nv) {
        double md = 0, s0 = 0, s1x = 0, s1y = 0, sxx = 0, sxy = 0, syy = 0;
        const size_t cs = (size_t)W.cap;
        const int w0 = g_vstart[v], w1 = g_vstart[v + 1];
        // two chunks per trip: all fourteen loads in flight together, added in chunk order (a sample has one to three chunks
        // as a rule: one round trip to memory instead of one per chunk)
        for (int w = w0; w < w1; w += 2) {
          const int wb = min(w + 1, w1 - 1);
          double pa[7], pb[7];
#pragma unroll
          for (int q = 0; q < 7; q++) { pa[q] = g_part[q * cs + w]; pb[q] = g_part[q * cs + wb]; }
          md += pa[0]; s0 += pa[1]; s1x += pa[2]; s1y += pa[3]; sxx += pa[4]; sxy += pa[5]; syy += pa[6];
          if (w + 1 < w1) { md += pb[0]; s0 += pb[1]; s1x += pb[2]; s1y += pb[3]; sxx += pb[4]; sxy += pb[5]; syy += pb[6]; }
        }
        const int m = (int)md;
        if (m >= 6) {  // :291
          float cx, cy;
          cx = g_samples[3 * v]; cy = g_samples[3 * v + 1];
          const double m1x = s1x / s0, m1y = s1y / s0;
          const double ux = (double)cx + m1x, uy = (double)cy + m1y;
          const double cxx = sxx / s0 - m1x * m1x, cyx = sxy / s0 - m1x * m1y, cyy = syy / s0 - m1y * m1y;
          double lmin, lmax, vmin[2], vmax[2];
          eig2(cxx, cyx, cyy, &lmin, &lmax, vmin, vmax);
          const double cond = fabs(lmax / lmin);  // :53
          const double det = lmax * lmin;         // :54
          valid = ((cond <= 10000) && (det > 0.00001) && lmin > 0 && lmax > 0) ? 1 : 0;  // :56
          if (valid) {
            if (vmin[0] * (0.0 - ux) + vmin[1] * (0.0 - uy) < 0) { vmin[0] = -vmin[0]; vmin[1] = -vmin[1]; }  // :59-61
            c.mean[0] = ux; c.mean[1] = uy;
            c.cov[0] = cxx; c.cov[1] = cyx; c.cov[2] = cyy;
            c.normal[0] = vmin[0]; c.normal[1] = vmin[1];
            c.orth[0] = vmax[0]; c.orth[1] = vmax[1];
            c.lambda_min = lmin; c.lambda_max = lmax;
            c.scale = log(1.0 + cond / 2);  // :57
            c.sum_intensity = s0; c.avg_intensity = s0 / m;
            c.nsamples = m; c.valid = 1;
          }
        }
      }
      int round_total;
      const int o = base + block_exclusive_scan<CFEAR_FEAT_BLOCK>(valid, W.red_i, &round_total);
      if (valid && o < cap_cells) {
        typedef __attribute__((address_space(1))) cfear_cell g_cell;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(1))) f64x2 g_f64x2;
        if (keep_cells) {  // (block-uniform)
          g_cell* gc = (g_cell*)S->cells + o;
          gc->mean[0] = c.mean[0]; gc->mean[1] = c.mean[1]; gc->cov[0] = c.cov[0]; gc->cov[1] = c.cov[1]; gc->cov[2] = c.cov[2];
          gc->normal[0] = c.normal[0]; gc->normal[1] = c.normal[1]; gc->orth[0] = c.orth[0]; gc->orth[1] = c.orth[1];
          gc->lambda_min = c.lambda_min; gc->lambda_max = c.lambda_max; gc->scale = c.scale;
          gc->sum_intensity = c.sum_intensity; gc->avg_intensity = c.avg_intensity; gc->nsamples = c.nsamples; gc->valid = c.valid;
        }
        g_f64* rc = (g_f64*)S->rcov + 3 * (size_t)o;
        rc[0] = c.cov[0]; rc[1] = c.cov[1]; rc[2] = c.cov[2];
        g_f32* mf = (g_f32*)S->mean_f;
        mf[2 * o] = (float)c.mean[0];
        mf[2 * o + 1] = (float)c.mean[1];
        const size_t cc = (size_t)cap_cells;
        g_f64* rs = (g_f64*)S->rsrc + o;
        rs[0] = c.mean[0]; rs[cc] = c.mean[1]; rs[2 * cc] = c.normal[0]; rs[3 * cc] = c.normal[1];
        rs[4 * cc] = (double)c.nsamples; rs[5 * cc] = c.scale;
        g_f64x2* rt = (g_f64x2*)(S->rtar + 8 * (size_t)o);
        rt[0] = f64x2{c.mean[0], c.mean[1]}; rt[1] = f64x2{c.normal[0], c.normal[1]};
        rt[2] = f64x2{(double)c.nsamples, c.scale};
      }
      base += round_total;
    }
    n_cells_out = base < cap_cells ? base : cap_cells;
    if (tid == 0) { S->n_cells = n_cells_out; if (base > cap_cells) S->status = CFEAR_ERR_CAPACITY; }  // more cells than the scan block holds (cfear_tune MAX_CELLS): the first cap_cells are kept
    __syncthreads();
  }
  if (pt) pt->mark();
  if (pt) pt->mark();
  cell_grid_block(S, n_cells_out, P, W, false, pt);
}

// GetClosestIdx (pointnormal.cpp:238-254): 1-NN over the float cell means, accepted iff d2 < d*d.
// Exact-distance ties resolve to the lowest cell index (same rule as the oracle).
// what the 1-NN search needs from a scan (the registration keeps one per keyframe in LDS: no pointer chasing)
struct GridView {
  const int* gs; const float4* gp; const double* rtar;
  double igc;  // 1 / bucket size, in double: the queries divide by it (a full division per query and keyframe otherwise)
  float gminx, gminy;
  int gw, gh, n_cells;
};
__device__ __forceinline__ GridView grid_view(const ScanDev* S) {
  GridView G;
  G.gs = S->gstart; G.gp = S->gpts; G.rtar = S->rtar; G.gminx = S->gminx; G.gminy = S->gminy; G.igc = 1.0 / (double)S->gcell;
  G.gw = S->gw; G.gh = S->gh; G.n_cells = S->n_cells;
  return G;
}
// TIE_HIGH: exact-distance ties to the HIGHEST cell index (cfear_tune NN_TIE_RULE = 1: the other end of the admissible answers, for sensitivity
// runs on the device like the oracle's CFO_PERT_NN_TIE_HIGH); the production rule is the lowest
template <bool TIE_HIGH = false>
__device__ inline int scan_closest(const GridView& S, double px, double py, double d) {
  const float qx = (float)px, qy = (float)py;
  const int gw = S.gw, gh = S.gh;
  if (S.n_cells <= 0 || gw <= 0) return -1;
  const double m = d * (1.0 + 1e-6) + 1e-6;
  // one reciprocal instead of four divisions: the window is padded by m - d >= 1e-6, an ulp in the bucket coordinate
  // cannot uncover anything within d of the query
  const double igc = S.igc, gmx = (double)S.gminx, gmy = (double)S.gminy;
  int gx0 = (int)floor(((double)qx - m - gmx) * igc), gx1 = (int)floor(((double)qx + m - gmx) * igc);
  int gy0 = (int)floor(((double)qy - m - gmy) * igc), gy1 = (int)floor(((double)qy + m - gmy) * igc);
  // the builder clamps bucket coordinates, so clamp the query window the same way
  gx0 = max(gx0, 0); gy0 = max(gy0, 0); gx1 = min(gx1, gw - 1); gy1 = min(gy1, gh - 1);
  if (gx0 > gx1 || gy0 > gy1) return -1;
  const int* __restrict__ gs = S.gs;
  const unsigned short* __restrict__ g16 = grid_off16(S.gs);
  const bool small = S.n_cells <= CFEAR_GRID16_MAX;  // 16-bit offsets (grid_off16), else 32-bit ones in gstart
  const float4* __restrict__ gp = S.gp;
  int best = -1;
  float bd = 3.4e38f;
  for (int gy = gy0; gy <= gy1; gy += 3) {  // three rows at a time: all bucket bounds in flight together
    int ra[3], rb[3];
#pragma unroll
    for (int r = 0; r < 3; r++) {
      const int row = min(gy + r, gy1);
      if (small) { ra[r] = (int)g16[row * gw + gx0]; rb[r] = (int)g16[row * gw + gx1 + 1]; }
      else { ra[r] = gs[row * gw + gx0]; rb[r] = gs[row * gw + gx1 + 1]; }
      if (gy + r > gy1) rb[r] = ra[r];
    }
#pragma unroll
    for (int r = 0; r < 3; r++) {
      for (int q = ra[r]; q < rb[r]; q += 2) {
        float4 c[2];
#pragma unroll
        for (int u = 0; u < 2; u++) c[u] = gp[min(q + u, rb[r] - 1)];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          const float dx = qx - c[u].x, dy = qy - c[u].y;
          float d2 = dx * dx; d2 += dy * dy;
          const int i = __float_as_int(c[u].z);
          if (q + u < rb[r] && (d2 < bd || (d2 == bd && (TIE_HIGH ? i > best : i < best)))) { bd = d2; best = i; }
        }
      }
    }
  }
  if (best >= 0 && (double)bd < d * d) return best;
  return -1;
}
// GetClosestIdx under a tie rule other than the production one (cfear_tune NN_TIE_RULE; parity / sensitivity modes, slow paths): 1 = highest
// index, 2 = what FLANN's kd-tree descent returns (kdtree_flann_dev.h; the scan must have been built in that mode). stack: CFEAR_KD_STACK
// entries of this thread's own
__device__ inline int scan_closest_rule(const ScanDev* S, const GridView& G, double px, double py, double d, int rule, KdVisit* stack) {
  if (rule == 2 && S->kd.nodes) {
    float bd = 0.f;
    const int best = kd_nearest(&S->kd, (float)px, (float)py, stack, &bd);  // kd_cells.nearestKSearch(pnt, 1, ...)
    return (best >= 0 && (double)bd < d * d) ? best : -1;                   // pointNKNSquaredDistances[0] < d*d
  }
  return rule == 1 ? scan_closest<true>(G, px, py, d) : scan_closest<false>(G, px, py, d);
}

}  // namespace cfear_dev
