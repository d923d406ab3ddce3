// cfar.hip -- azimuth CA-CFAR, the alternative stage-1 filter of radarDriver::Process
// (radar_driver.cpp:52-56: AzimuthCACFAR(window_size, false_alarm_rate, nb_guard_cells, range_res, z_min,
// min_distance, 400.0).getFilteredPointCloud, cfar.cpp:27-87).
//
// Per range bin that passes the static test, the detector compares the squared intensity with a scaled mean of the
// squared intensities in a trailing and a forwarding window (guard cells in between). Three launches per batch:
//   cfar_detect_kernel  one 256-thread workgroup per azimuth row: the row is staged in LDS with aligned dword loads, the windows
//                       become differences of an LDS prefix sum of squares (integers, exact); the image is read ONCE and what comes
//                       out is a bit per range bin (the row's hit mask) and the row's count;
//   cfar_row_scan_kernel  row counts -> row offsets of every image (the output cloud is row-major over (azimuth, range bin) like
//                       the reference's push_back order);
//   cfar_emit_kernel    a wave per four rows walks their masks and writes the points (the intensity is a gather of the hit bytes).
// The decision replays the reference's double arithmetic (sum / N per window, (t + f) / 2, scaling * mean, I^2 > threshold; an empty
// window gives 0/0 = NaN and no detection) - but only where it has to: with z_min = 20 (the reference's own CA-CFAR preset,
// params/kstrong_vs_cfar/oxford-cfear-3-ca-cfar:27) nearly every bin passes the static test, and two double divisions per bin made
// the detector 40 x slower than the whole k-strongest pipeline. For a bin whose two windows are complete the threshold is
// scaling / (2 w) * (sum of both windows), a u32 times a constant: it is formed in float first (relative error < 2e-7), and only a bin
// whose I^2 lies within 2e-6 of that threshold - or whose windows are clipped by the row's ends - goes through the reference's
// expressions in double. The result is the same bit for bit (tests/test_cfar_gpu.py); LDS is sized by the row length (17 KB at 3360
// bins: nine rows in flight per compute unit instead of one).
#include <math.h>
#include <stdlib.h>

#include "blockops.h"
#include "common.h"

namespace {
using namespace cfear_dev;

constexpr int CFAR_BLOCK = 256;
constexpr int CFAR_MAX_R = 16384;  // range bins per azimuth the LDS row / prefix arrays hold

struct CfarParams {
  int A, R, window, guard;
  int ilo, ihi;     // bins that pass the range test (cfar.cpp:45: range > min_distance && range < max_distance), found on the host with the
                    // reference's own double expressions: ilo <= i <= ihi
  int iv_min;       // smallest intensity with (double)I > static_threshold (256: none)
  int mask_words;   // 32-bit words of a row's hit mask (even: a wave's ballot is two words)
  int ipt;          // consecutive bins per thread of the prefix pass (odd: conflict-free LDS strides)
  float kf;         // scaling / (2 w) in float
  double range_res, scaling;
  // owner-layout detector (cfar_detect_owner_kernel): prefix sums are kept as P << sh, a bin is worth a closer look when (S << sh) - I^2 * kopen <= 0
  int sh, kopen;
  int cr[4], cq[4]; // owner-layout detector, set per launch shape: window bound c of bin TB t + e is row e + cr[c], column t + cq[c] (off = TB cq + cr)
  int rmin, rmax;   // ... rows rmin .. TB + rmax - 1 exist (rmin <= 0 <= rmax)
  int stop;         // profiling (CFEAR_CFAR_STOP = n): a row's trip ends after phase n (1 prefix, 2 integer test); 0 = the product
};

// the reference's decision as written (cfar.cpp:47-60), windows clipped by the row's ends
__device__ __noinline__ bool cfar_exact(const uint32_t* prefix, int i, int iv, int R, int guard, int window, double scaling) {
  const int t0 = max(0, i - guard - window), t1 = i - guard;                           // :48-49
  const int f0 = i + guard, f1 = min(R, i + guard + window);                           // :52-53
  const double tn = t1 > t0 ? (double)(t1 - t0) : 0.0, fn = f1 > f0 ? (double)(f1 - f0) : 0.0;
  const double ts = t1 > t0 ? (double)(prefix[t1] - prefix[t0]) : 0.0;
  const double fs = f1 > f0 ? (double)(prefix[f1] - prefix[f0]) : 0.0;
  const double mean = (ts / tn + fs / fn) / 2.0;  // empty window: 0/0 = NaN -> no detection (:56)
  const double threshold = scaling * mean;
  return (double)(iv * iv) > threshold;                                                // :58-60
}

__global__ __launch_bounds__(CFAR_BLOCK) void cfar_detect_kernel(const uint8_t* __restrict__ polar, long long alloc_bytes, CfarParams P,
                                                                 int* __restrict__ row_count, uint32_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) uint32_t cfar_lds[];  // prefix[R + 1] (prefix[i] = sum of squares of bins < i), then the row's bytes
  __shared__ int red_i[64];
  const int R = P.R, tid = threadIdx.x;
  uint32_t* prefix = cfar_lds;
  uint8_t* rowbuf = reinterpret_cast<uint8_t*>(cfar_lds + (R + 1));
  // the images lie back to back, so row blockIdx.x starts at blockIdx.x * R
  const int grow = blockIdx.x;
  // ---- stage the row with aligned dword loads (the bytes around the row belong to the neighbouring rows) ----
  const long long row_off = (long long)grow * R;
  const int first = (int)(row_off & 3);
  const long long base = row_off - first;
  const int ndw = (first + R + 3) >> 2;
  for (int i = tid; i < ndw; i += CFAR_BLOCK) {
    const long long o = base + 4LL * i;
    uint32_t v;
    if (o + 4 <= alloc_bytes) {
      v = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(polar + o));
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
  {
    const int b0 = min(R, tid * P.ipt), b1 = min(R, b0 + P.ipt);
    int s = 0;
    for (int i = b0; i < b1; i++) { const int v = row[i]; s += v * v; }
    int tot;
    int o = block_exclusive_scan<CFAR_BLOCK>(s, red_i, &tot);
    for (int i = b0; i < b1; i++) { prefix[i] = (uint32_t)o; const int v = row[i]; o += v * v; }
    if (tid == 0) prefix[R] = (uint32_t)tot;
    __syncthreads();
  }
  // ---- decisions: thread <-> bin, 64 consecutive bins per wave and trip (the ballot is the mask) ----
  const int lane = tid & 63, g = P.guard, w = P.window;
  uint32_t* mrow = mask + (size_t)grow * P.mask_words;
  int cnt = 0;  // lane 0 of every wave: hits of the wave's trips
  for (int i0 = (tid >> 6) * 64; i0 < R; i0 += CFAR_BLOCK) {
    const int i = i0 + lane;
    const bool in = i < R && i >= P.ilo && i <= P.ihi;
    const int iv = row[i < R ? i : R - 1];
    const bool cand = in && iv >= P.iv_min;  // cfar.cpp:45
    const int t0 = i - g - w, f1 = i + g + w;
    const bool interior = t0 >= 0 && f1 <= R;  // both windows complete: N = w each
    bool hit = false, unsure = cand && !interior;
    if (cand && interior) {
      const uint32_t S = (prefix[i - g] - prefix[t0]) + (prefix[f1] - prefix[i + g]);
      const float thr = (float)S * P.kf, I2 = (float)(iv * iv);
      hit = I2 > thr * 1.000002f;
      unsure = !hit && I2 >= thr * 0.999998f;
    }
    if (unsure) hit = cfar_exact(prefix, i, iv, R, g, w, P.scaling);
    const unsigned long long m = __ballot(hit);
    if (lane == 0) {
      mrow[i0 >> 5] = (uint32_t)m; mrow[(i0 >> 5) + 1] = (uint32_t)(m >> 32);
      cnt += __popcll(m);
    }
  }
  if (lane == 0) red_i[32 + (tid >> 6)] = cnt;
  __syncthreads();
  if (tid == 0) row_count[grow] = red_i[32] + red_i[33] + red_i[34] + red_i[35];
}

// ---- the same detector with a third of the instructions (rows whose length is a multiple of four, up to 256 * TB bins: every radar here) --------
// The kernel above is bound by vector instruction issue (PMC: 2960 vector + 1820 scalar instructions per row, SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES x
// nine waves per SIMD = 1.05). Same shape here - one 256-thread workgroup per row, nine rows in flight per unit - with the work cut down:
//   * the row is staged as dwords and thread t owns the bins [TB t, TB t + TB): their sum of squares is TB / 4 dot products (v_dot4_u32_u8 of a dword
//     with itself), one block scan gives the segment's offset, the per-bin prefix goes out as 16-byte LDS stores;
//   * the prefix array is padded by guard + window entries on both sides - zeros in front, the row total behind - so that a window clipped by a row
//     end is simply the difference of two entries: no address clamps, and the exact path of such a bin needs arithmetic only (no memory);
//   * a wave takes four trips of 64 consecutive bins at a time: sixteen prefix reads off four address registers (constant offsets), and in the
//     middle of the row - every bin inside the range gate, both windows complete - three comparisons per bin; the comparison's result mask IS the
//     hit mask, counts stay on the scalar unit.
// Same decisions as the kernel above (the float pre-test only ever hands doubtful bins to the reference's double arithmetic): tests/test_cfar_gpu.py.
__device__ __noinline__ bool cfar_exact_vals(uint32_t ts_u, int tn_i, uint32_t fs_u, int fn_i, int iv2, double scaling) {
  const double tn = tn_i > 0 ? (double)tn_i : 0.0, fn = fn_i > 0 ? (double)fn_i : 0.0;
  const double ts = tn_i > 0 ? (double)ts_u : 0.0, fs = fn_i > 0 ? (double)fs_u : 0.0;
  const double mean = (ts / tn + fs / fn) / 2.0;  // empty window: 0/0 = NaN -> no detection (cfar.cpp:56)
  const double threshold = scaling * mean;
  return (double)iv2 > threshold;                 // :58-60
}
__device__ __forceinline__ uint32_t lane_write(uint32_t acc, uint32_t sval, const int lane_const) {  // acc's lane lane_const := a scalar (one instruction)
  asm volatile("v_writelane_b32 %0, %1, %2" : "+v"(acc) : "s"(sval), "n"(lane_const));
  return acc;
}
template <int KI, int TB>
__device__ __forceinline__ void cfar_four_trips(const CfarParams& P, __attribute__((address_space(3))) uint32_t* prefix, const int trip0, const int ntrips,
                                                const int lane, const float kf_lo, const float kf_hi, const float kscale, const int iv_min2, uint32_t& acc, int& cnt) {
  typedef __attribute__((address_space(3))) uint32_t l_u32;
  constexpr int U = 4;
  const int R = P.R, g = P.guard, w = P.window;
  uint32_t S[U]; int iv2[U]; uint32_t ts[U], fs[U];
  const int ia = trip0 * 64, ib = ia + 64 * U - 1;  // the bins of these trips (scalars: trip0 is)
  {  // twenty reads off five address registers (constant offsets): I^2 is a difference of the prefix too
    l_u32* const pa = prefix + (ia + lane - g - w); l_u32* const pb = pa + w; l_u32* const pc = pb + 2 * g; l_u32* const pd = pc + w;
    l_u32* const pi = prefix + (ia + lane);
#pragma unroll
    for (int u = 0; u < U; u++) {
      ts[u] = pb[64 * u] - pa[64 * u]; fs[u] = pd[64 * u] - pc[64 * u];
      S[u] = ts[u] + fs[u];
      iv2[u] = (int)(pi[64 * u + 1] - pi[64 * u]);
    }
  }
  // (scalar) every bin inside the range gate, both windows complete: the middle of the row
  if (ia - g - w >= 0 && ib + g + w <= R && ia >= P.ilo && ib <= P.ihi) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      unsigned long long m = 0;
      // (every predicate a comparison of its own, combined on the scalar unit: a ballot of `a && b` comes back through a VGPR)
      const unsigned long long candm = __builtin_amdgcn_ballot_w64(iv2[u] >= iv_min2);  // cfar.cpp:45 (I >= I_min <=> I^2 >= I_min^2)
      if (candm) {
        const float Sf = (float)S[u], I2 = (float)iv2[u];
        const unsigned long long open = candm & __builtin_amdgcn_ballot_w64(I2 >= Sf * kf_lo);  // hits are one bin in a few hundred: the common trip ends here
        if (open) {
          const unsigned long long surem = __builtin_amdgcn_ballot_w64(I2 > Sf * kf_hi);
          m = open & surem;
          const unsigned long long tiem = open & ~surem;  // near-ties of the float test: the reference's arithmetic
          if (tiem) {
            bool hit = false;
            if ((tiem >> lane) & 1) hit = cfar_exact_vals(ts[u], w, fs[u], w, iv2[u], P.scaling);
            m |= __builtin_amdgcn_ballot_w64(hit);
          }
        }
      }
      cnt += __popcll(m);
      acc = lane_write(acc, (uint32_t)m, 8 * KI + 2 * u);
      acc = lane_write(acc, (uint32_t)(m >> 32), 8 * KI + 2 * u + 1);
    }
  } else {
#pragma unroll
    for (int u = 0; u < U; u++) {
      unsigned long long m = 0;
      const int i = ia + 64 * u + lane;
      const bool cand = trip0 + u < ntrips && iv2[u] >= iv_min2 && i >= P.ilo && i <= P.ihi;
      if (__builtin_amdgcn_ballot_w64(cand)) {
        // a window clipped by a row end: the mean of two means; in float first (a relative 1e-5 covers its handful of roundings), the reference's
        // double arithmetic for what that cannot decide. An empty window makes the reference's mean NaN: no detection.
        const int tn = min(i - g, w), fn = min(R - i - g, w);
        const float thr = kscale * ((float)ts[u] * __builtin_amdgcn_rcpf((float)tn) + (float)fs[u] * __builtin_amdgcn_rcpf((float)fn));
        const float I2 = (float)iv2[u];
        const bool both = tn > 0 && fn > 0;
        const bool sure = both && I2 > thr * 1.00001f;
        bool hit = cand && sure;
        const bool doubt = cand && both && !sure && I2 >= thr * 0.99999f;
        if (__builtin_amdgcn_ballot_w64(doubt)) {
          if (doubt) hit = cfar_exact_vals(ts[u], tn, fs[u], fn, iv2[u], P.scaling);
        }
        m = __builtin_amdgcn_ballot_w64(hit);
      }
      cnt += __popcll(m);
      acc = lane_write(acc, (uint32_t)m, 8 * KI + 2 * u);
      acc = lane_write(acc, (uint32_t)(m >> 32), 8 * KI + 2 * u + 1);
    }
  }
}
template <int TB /* bins per thread of the prefix pass: 16 or 32 */>
__global__ __launch_bounds__(CFAR_BLOCK) void cfar_detect_fast_kernel(const uint8_t* __restrict__ polar, CfarParams P, int pad, int* __restrict__ row_count,
                                                                      uint32_t* __restrict__ mask) {
  extern __shared__ __attribute__((aligned(16))) uint32_t cfar_lds[];  // prefix[-pad .. 256 TB + pad] (a bin's own square is a difference of it too)
  __shared__ int red_i[64];
  typedef __attribute__((address_space(3))) uint32_t l_u32;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
  typedef __attribute__((address_space(3))) u32x4 l_u32x4;
  typedef __attribute__((address_space(1))) const uint32_t g_cu32;
  typedef __attribute__((address_space(1))) const u32x4a g_cu32x4;
  constexpr int ND = TB / 4, NB = CFAR_BLOCK * TB;  // dwords per thread, bins the segments cover
  const int R = P.R, tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  l_u32* const prefix = (l_u32*)cfar_lds + pad;
  const int grow = blockIdx.x, ndw = R >> 2;
  g_cu32* src = (g_cu32*)(polar + (long long)grow * R);
  uint32_t seg[ND];
  if (ND * tid + ND <= ndw) {  // this thread's own bins, straight from memory: 16 bytes per load, consecutive threads consecutive
#pragma unroll
    for (int j = 0; j < ND; j += 4) {
      const u32x4 v = __builtin_nontemporal_load((g_cu32x4*)(src + ND * tid + j));
      seg[j] = v.x; seg[j + 1] = v.y; seg[j + 2] = v.z; seg[j + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int j = 0; j < ND; j++) seg[j] = ND * tid + j < ndw ? __builtin_nontemporal_load(src + ND * tid + j) : 0u;
  }
  uint32_t s = 0;
#pragma unroll
  for (int j = 0; j < ND; j++) s = __builtin_amdgcn_udot4(seg[j], seg[j], s, false);
  for (int i = tid; i < pad; i += CFAR_BLOCK) prefix[-pad + i] = 0u;
  int tot;
  uint32_t o = (uint32_t)block_exclusive_scan<CFAR_BLOCK>((int)s, red_i, &tot);
#pragma unroll
  for (int j = 0; j < ND; j++) {
    const uint32_t d = seg[j], b0 = d & 0xFFu, b1 = (d >> 8) & 0xFFu, b2 = (d >> 16) & 0xFFu, b3 = d >> 24;
    u32x4 v;
    v.x = o; o += b0 * b0; v.y = o; o += b1 * b1; v.z = o; o += b2 * b2; v.w = o; o += b3 * b3;
    *(l_u32x4*)(prefix + TB * tid + 4 * j) = v;  // prefix[i .. i + 3], i = TB tid + 4 j (bins >= R are zeros: prefix[R ..] = the row total)
  }
  for (int i = NB + tid; i <= NB + pad; i += CFAR_BLOCK) prefix[i] = (uint32_t)tot;
  __syncthreads();
  // ---- decisions: wave wv takes the iterations wv, wv + 4, ... of four trips (256 bins) each; their masks collect in the lanes of one register ----
  const int ntrips = (R + 63) >> 6;
  const float kf_hi = P.kf * 1.000002f, kf_lo = P.kf * 0.999998f, kscale = P.kf * (float)P.window;  // (scaling / 2, to a rounding: the margins cover it)
  const int iv_min2 = P.iv_min * P.iv_min;
  constexpr int ITERS = TB / 4;  // NB / 256 iterations over four waves
  int cnt = 0;
  uint32_t acc = 0;
  if (4 * wv < ntrips) cfar_four_trips<0, TB>(P, prefix, 4 * wv, ntrips, lane, kf_lo, kf_hi, kscale, iv_min2, acc, cnt);
  if (4 * (wv + 4) < ntrips) cfar_four_trips<1, TB>(P, prefix, 4 * (wv + 4), ntrips, lane, kf_lo, kf_hi, kscale, iv_min2, acc, cnt);
  if (4 * (wv + 8) < ntrips) cfar_four_trips<2, TB>(P, prefix, 4 * (wv + 8), ntrips, lane, kf_lo, kf_hi, kscale, iv_min2, acc, cnt);
  if (4 * (wv + 12) < ntrips) cfar_four_trips<3, TB>(P, prefix, 4 * (wv + 12), ntrips, lane, kf_lo, kf_hi, kscale, iv_min2, acc, cnt);
  if constexpr (ITERS > 4) {
    if (4 * (wv + 16) < ntrips) cfar_four_trips<4, TB>(P, prefix, 4 * (wv + 16), ntrips, lane, kf_lo, kf_hi, kscale, iv_min2, acc, cnt);
    if (4 * (wv + 20) < ntrips) cfar_four_trips<5, TB>(P, prefix, 4 * (wv + 20), ntrips, lane, kf_lo, kf_hi, kscale, iv_min2, acc, cnt);
    if (4 * (wv + 24) < ntrips) cfar_four_trips<6, TB>(P, prefix, 4 * (wv + 24), ntrips, lane, kf_lo, kf_hi, kscale, iv_min2, acc, cnt);
    if (4 * (wv + 28) < ntrips) cfar_four_trips<7, TB>(P, prefix, 4 * (wv + 28), ntrips, lane, kf_lo, kf_hi, kscale, iv_min2, acc, cnt);
  }
  {  // lane 8 k + 2 u + h holds half h of trip 4 (wv + 4 k) + u
    const int trip = 4 * (wv + 4 * (lane >> 3)) + ((lane >> 1) & 3);
    if (lane < 8 * ITERS && trip < ntrips) mask[(size_t)grow * P.mask_words + 2 * trip + (lane & 1)] = acc;
  }
  if (lane == 0) red_i[32 + wv] = cnt;
  __syncthreads();
  if (tid == 0) row_count[grow] = red_i[32] + red_i[33] + red_i[34] + red_i[35];
}


// ---- round 6: the detector with a thread owning TB consecutive bins from the load to the decision -----------------------------------------
// cfar_detect_fast_kernel issues 1477 vector + 810 scalar + 200 LDS instructions per row (0.88 of the vector issue slots, 0.13 of the HBM rate): a
// bin's decision is made by another lane than the one that squared it, I^2 comes back out of LDS, every comparison is a scalar mask of its own, and
// the hit mask goes to memory for a second kernel that fetches the image again. Here
//   * thread t keeps its TB bins' squares q[e] in registers; the prefix sum goes to LDS already scaled, P' = P << sh, in a transposed layout
//     [e][t] (row e = the e-th bin of every thread: consecutive lanes, consecutive banks), every row twice - row TB + e holds row e one thread
//     further - so that bin TB t + e + off is (row e + off mod TB, column t + off div TB) for EVERY e: the four window bounds of a bin are four
//     ds_read_b32 off four address registers with compile-time offsets, no address arithmetic, no bank conflicts;
//   * the test is in integers: I^2 > kf S  <=  (S << sh) - I^2 * kopen <= 0 with kopen = ceil((1 + 4e-6) 2^sh / kf) (v_mad_i32_i24 on the square
//     the thread still holds; the sign bits collect in one register with v_alignbit). That is a superset of the hits by construction; the few bins
//     it leaves (nine per row are hits, a fraction of that near-misses) go through the float pre-test and the reference's double arithmetic
//     exactly as in the kernels above - same decisions, bit for bit (tests/test_cfar_gpu.py);
//   * bins whose windows are clipped by a row end (g + w bins at either end, where the range gate lets them through) are decided by a short
//     loop over just those bins with the general expressions and OR-ed into their owners' masks through LDS;
//   * what leaves the kernel is the hits themselves - (bin, intensity) records in bin order per (row, wave) segment, and the segment's count
//     - not a bit per bin: the second kernel (cfar_emit_recs_kernel) never touches the image or a mask. A segment with more hits than its slot
//     holds (CFAR_SEG_CAP) writes its threads' hit masks instead and the emit kernel walks those (rare: a row of clutter).
// The workgroups are persistent (a row per trip, the next row's loads in flight during the current row's decisions).
constexpr int CFAR_SEG_CAP = 64;  // records per (row, wave) segment
constexpr int CFAR_SEG_MASKS = 1 << 30;  // flag on a segment's count: its hits are in the threads' masks, not in records

// thread's ND dwords of a row. CLAMPED: the caller hands a pointer inside the row for every thread (threads past the row's end one that they
// then zero: cfar_zero_segment) - no branch around the loads, so the outstanding requests stay countable (a wait for the oldest of several rows in
// flight instead of a wait for all of them).
template <int ND>
__device__ __forceinline__ void cfar_load_segment_full(__attribute__((address_space(1))) const uint32_t* p, uint32_t (&seg)[ND]) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4a __attribute__((ext_vector_type(4), aligned(4)));
  typedef uint32_t u32x2a __attribute__((ext_vector_type(2), aligned(4)));
  typedef __attribute__((address_space(1))) const u32x4a g_cu32x4;
  typedef __attribute__((address_space(1))) const u32x2a g_cu32x2;
  constexpr int N4 = ND / 4 * 4, N2 = ND / 2 * 2;
#pragma unroll
  for (int j = 0; j < N4; j += 4) {
    const u32x4 v = __builtin_nontemporal_load((g_cu32x4*)(p + j));
    seg[j] = v.x; seg[j + 1] = v.y; seg[j + 2] = v.z; seg[j + 3] = v.w;
  }
  if constexpr (N2 > N4) {
    const u32x2 v = __builtin_nontemporal_load((g_cu32x2*)(p + N4));
    seg[N4] = v.x; seg[N4 + 1] = v.y;
  }
  if constexpr (ND > N2) seg[N2] = __builtin_nontemporal_load(p + N2);
}
template <int ND>
__device__ __forceinline__ void cfar_load_segment(__attribute__((address_space(1))) const uint32_t* p, int navail, uint32_t (&seg)[ND]) {
  if (navail >= ND) {
    cfar_load_segment_full<ND>(p, seg);
  } else {
#pragma unroll
    for (int j = 0; j < ND; j++) seg[j] = j < navail ? __builtin_nontemporal_load(p + j) : 0u;
  }
}

template <int TB, int NW, int RS /* columns of the transposed prefix array: a multiple of 64, so that the reads of two rows pair up as ds_read2st64_b32 off one base */,
          bool FULL /* the row is a whole number of threads' segments: branch-free loads, several rows in flight */>
__global__ __launch_bounds__(64 * NW, NW == 2 ? 3 : 1) void cfar_detect_owner_kernel(const uint8_t* __restrict__ polar, CfarParams P, int rows, int padl, int* __restrict__ seg_count,
                                                                    uint32_t* __restrict__ recs, uint32_t* __restrict__ hmask) {
  constexpr int NT = 64 * NW, ND = TB / 4;
  static_assert(TB % 4 == 0 && TB <= 32 && RS % 64 == 0, "bins per thread / row stride");
  extern __shared__ __attribute__((aligned(16))) uint32_t cfar_lds[];
  typedef __attribute__((address_space(3))) uint32_t l_u32;
  typedef __attribute__((address_space(1))) const uint32_t g_cu32;
  // [nrows][RS] scaled prefix sums: row rho of column padl + t holds P'[TB t + rho], rho = rmin .. TB + rmax - 1 - the thread's own TB values and,
  // either side, copies of its neighbours' first rmax / last -rmin ones, so that bin TB t + e + off is (row e + r, column t + q) for every e with ONE
  // (r, q) per window bound: off = TB q + r with r taken in (-TB, TB) such that the rows all four bounds need are as few as possible (the host's choice:
  // for guard 10, window 40 and 28 bins per thread r = 6, -10, 10, -6: 48 rows instead of the 56 of r in [0, TB))
  const int nrows = TB + P.rmax - P.rmin;
  l_u32* const lp = (l_u32*)cfar_lds;                 // physical row 0 = row rmin
  l_u32* const lp0 = lp - P.rmin * RS;                // row 0
  l_u32* const lcand = lp + nrows * RS;               // [NW][64] a wave's candidate bins (what the integer test left)
  l_u32* const lstage = lcand + NW * 64;              // [NW][CFAR_SEG_CAP] a wave's records of the row, until the next trip writes them out
  int* const red_i = (int*)(lstage + NW * CFAR_SEG_CAP);  // [16] scan scratch
  const int R = P.R, g = P.guard, w = P.window, sh = P.sh, ndw = R >> 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  l_u32* const pA = lp0 + (P.cr[0] * RS + padl + tid + P.cq[0]);
  l_u32* const pB = lp0 + (P.cr[1] * RS + padl + tid + P.cq[1]);
  l_u32* const pC = lp0 + (P.cr[2] * RS + padl + tid + P.cq[2]);
  l_u32* const pD = lp0 + (P.cr[3] * RS + padl + tid + P.cq[3]);
  l_u32* const pown = lp0 + (padl + tid);  // P'[TB t + e] = pown[e RS]
  // Columns: thread t's at padl + t. Left of thread 0 the array reads 0 (bins < 0: written once, here), right of the thread that holds bin R it
  // reads the row total (bins >= R): a window a row end clips is then simply the difference of two entries, like everywhere else. The threads past
  // the row's end write those columns as a matter of course (their bytes are zeros); where the workgroup has no such thread a loop fills them in.
  const int tR = R / TB, padr = (g + w + TB - 1) / TB + 1;
  const int last_col_thread = min(tR + padr, RS - 1 - padl);
  const bool writes = tid <= last_col_thread;
  for (int k = tid; k < nrows * (padl + 1); k += NT) lp[(k / (padl + 1)) * RS + k % (padl + 1)] = 0u;  // (column padl: the rows below 0 are thread -1's; thread 0 writes the others every row)
  // this thread's bins inside the range gate (cfar.cpp:45)
  uint32_t vmask;
  {
    int lo_e = P.ilo - TB * tid, hi_e = P.ihi + 1 - TB * tid;
    lo_e = lo_e < 0 ? 0 : (lo_e > TB ? TB : lo_e);
    hi_e = hi_e < 0 ? 0 : (hi_e > TB ? TB : hi_e);
    vmask = hi_e > lo_e ? ((hi_e >= 32 ? 0xFFFFFFFFu : ((1u << hi_e) - 1u)) & ~((1u << lo_e) - 1u)) : 0u;
  }
  const int navail = ndw - ND * tid;  // dwords of the row from this thread's first on
  const int nK = -P.kopen;
  const float kscale = P.kf * (float)w;  // scaling / 2, to a rounding (the margins of the pre-test cover it)
  const int iv_min2 = P.iv_min * P.iv_min;
  const uint32_t scale = 1u << sh;
  // The decision for one bin the integer test could not rule out (any bin of the gate): the window sums off the prefix array, a float pre-test, the
  // reference's own arithmetic for what that cannot decide (cfar.cpp:47-60). The integer test is a superset test for the clipped bins too: their
  // windows are shorter than w, so their means - sums over fewer bins - are at least the sums over w.
  auto decide = [&](int bin, int* iv2_out) -> bool {
    const int t = bin / TB, e = bin - t * TB, dt = t - tid;
    const uint32_t ts = (pB[e * RS + dt] - pA[e * RS + dt]) >> sh, fs = (pD[e * RS + dt] - pC[e * RS + dt]) >> sh;
    const uint32_t p1 = e + 1 < TB ? pown[(e + 1) * RS + dt] : pown[dt + 1];  // P'[bin + 1]: the next row, or the next thread's first
    const int iv2 = (int)((p1 - pown[e * RS + dt]) >> sh);
    *iv2_out = iv2;
    const int t1 = bin - g, f0 = bin + g;
    const int tn = t1 > 0 ? min(t1, w) : 0, fn = f0 < R ? min(R - f0, w) : 0;  // cfar.cpp:48-53
    if (!(iv2 >= iv_min2 && tn > 0 && fn > 0)) return false;  // :45; an empty window makes the reference's mean NaN: no detection (:56)
    const float thr = kscale * ((float)ts * __builtin_amdgcn_rcpf((float)tn) + (float)fs * __builtin_amdgcn_rcpf((float)fn));
    const float I2 = (float)iv2;
    if (I2 > thr * 1.00001f) return true;
    if (!(I2 >= thr * 0.99999f)) return false;
    return cfar_exact_vals(ts, tn, fs, fn, iv2, P.scaling);
  };

  // Rows in flight: the memory system needs ~2 us of requests outstanding to stream at its rate, and a compute unit holds five of these workgroups,
  // so every workgroup keeps the next PF rows' bytes on their way (registers: ND each) while it works on the current one. (Rows past the end of the
  // batch re-read the last row: nothing looks at them.)
  constexpr int NBUF = 3;  // rows in flight per workgroup, the current one included
  uint32_t buf0[ND], buf1[ND], buf2[ND];
  int grow = blockIdx.x;
  const bool has_bytes = navail >= ND;           // FULL: every thread holds a whole segment or none
  const int toff = has_bytes ? ND * tid : 0;     // (threads without bytes read thread 0's and zero them)
  auto request = [&](long long gk, uint32_t (&dst)[ND]) __attribute__((always_inline)) {
    if constexpr (FULL) {
      if (gk > rows - 1) gk = rows - 1;
      cfar_load_segment_full<ND>((g_cu32*)(polar + gk * R) + toff, dst);
    } else {
      if (gk < rows) cfar_load_segment<ND>((g_cu32*)(polar + gk * R) + ND * tid, navail, dst);
    }
  };
  request(grow, buf0);
  request((long long)grow + gridDim.x, buf1);
  request((long long)grow + 2LL * gridDim.x, buf2);
  int prev_total = -1;  // records of the previous trip's row waiting in lstage (-1: nothing to write)
  int prev_row = 0;
  // one row; `seg` holds its bytes and is asked for the row NBUF trips ahead as soon as its squares are taken. (The three register sets take turns -
  // the loop below is unrolled by three - because moving a set on would mean waiting for it.)
  auto trip = [&](uint32_t (&seg)[ND]) __attribute__((always_inline)) {
    // ---- squares, the segment's sum, its offset in the row ----
    int nq[TB];       // -(I^2 kopen) - 1 per bin: what the integer test adds the window sums to
    uint32_t qs[TB];  // I^2 << sh
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < ND; j++) {
      const uint32_t d = (FULL && !has_bytes) ? 0u : seg[j], b[4] = {d & 0xFFu, (d >> 8) & 0xFFu, (d >> 16) & 0xFFu, d >> 24};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t q = __umul24(b[k], b[k]);
        nq[4 * j + k] = __mul24((int)q, nK) - 1;
        qs[4 * j + k] = __umul24(q, scale);
      }
      s = __builtin_amdgcn_udot4(d, d, s, false);
    }
#pragma unroll
    for (int e = 0; e < TB; e++) asm volatile("" : "+v"(nq[e]), "+v"(qs[e]));  // (the squares are taken before the registers they come from are loaded again)
    __builtin_amdgcn_sched_barrier(0);
    // The previous row's records leave now - after the wait for this row's bytes, not before it: one counter serves loads and stores, and a store
    // counts until the L2 has it - and the bytes of the row NBUF trips ahead are asked for right after.
    if (prev_total >= 0) {
      const size_t sidx = (size_t)prev_row * NW + wv;
      if (lane < prev_total && !(prev_total & CFAR_SEG_MASKS)) recs[sidx * CFAR_SEG_CAP + lane] = lstage[wv * CFAR_SEG_CAP + lane];
      if (lane == 0) seg_count[sidx] = prev_total;
    }
    request((long long)grow + (long long)NBUF * gridDim.x, seg);
    if (P.stop == 11) { prev_total = (nq[0] ^ nq[TB - 1] ^ (int)qs[3] ^ (int)s) == 0x12345 ? 0 : -1; return; }
    int tot;
    uint32_t o = (uint32_t)block_exclusive_scan_1b<NT>((int)s, red_i, 0, &tot) << sh;
    if (P.stop == 12) { prev_total = (nq[0] ^ nq[TB - 1] ^ (int)qs[3] ^ (int)o) == 0x12345 ? 0 : -1; __syncthreads(); return; }
    // (the barrier of the scan also ends the previous row's reads of everything written below)
    if (writes) {
      uint32_t (&pv)[TB] = qs;  // the squares' registers become the prefix values
#pragma unroll
      for (int e = 0; e < TB; e++) { const uint32_t t = qs[e]; pv[e] = o; pown[e * RS] = o; o += t; }
      // the copies for the neighbours: the first rmax values one column to the left in the rows above TB, the last -rmin one column to the right in
      // the rows below 0 (one jump into a run of stores each: the counts are the launch's, not the compiler's)
#define CFAR_DUP_HI(E) case (E) + 1: if constexpr ((E) < TB) pown[((E) + TB) * RS - 1] = pv[(E) < TB ? (E) : 0]; [[fallthrough]];
      switch (P.rmax) {
        CFAR_DUP_HI(31) CFAR_DUP_HI(30) CFAR_DUP_HI(29) CFAR_DUP_HI(28) CFAR_DUP_HI(27) CFAR_DUP_HI(26) CFAR_DUP_HI(25) CFAR_DUP_HI(24)
        CFAR_DUP_HI(23) CFAR_DUP_HI(22) CFAR_DUP_HI(21) CFAR_DUP_HI(20) CFAR_DUP_HI(19) CFAR_DUP_HI(18) CFAR_DUP_HI(17) CFAR_DUP_HI(16)
        CFAR_DUP_HI(15) CFAR_DUP_HI(14) CFAR_DUP_HI(13) CFAR_DUP_HI(12) CFAR_DUP_HI(11) CFAR_DUP_HI(10) CFAR_DUP_HI(9) CFAR_DUP_HI(8)
        CFAR_DUP_HI(7) CFAR_DUP_HI(6) CFAR_DUP_HI(5) CFAR_DUP_HI(4) CFAR_DUP_HI(3) CFAR_DUP_HI(2) CFAR_DUP_HI(1) CFAR_DUP_HI(0)
        default: break;
      }
#undef CFAR_DUP_HI
#define CFAR_DUP_LO(K) case (K): if constexpr ((K) <= TB) pown[-(K) * RS + 1] = pv[(K) <= TB ? TB - (K) : 0]; [[fallthrough]];
      switch (-P.rmin) {
        CFAR_DUP_LO(32) CFAR_DUP_LO(31) CFAR_DUP_LO(30) CFAR_DUP_LO(29) CFAR_DUP_LO(28) CFAR_DUP_LO(27) CFAR_DUP_LO(26) CFAR_DUP_LO(25)
        CFAR_DUP_LO(24) CFAR_DUP_LO(23) CFAR_DUP_LO(22) CFAR_DUP_LO(21) CFAR_DUP_LO(20) CFAR_DUP_LO(19) CFAR_DUP_LO(18) CFAR_DUP_LO(17)
        CFAR_DUP_LO(16) CFAR_DUP_LO(15) CFAR_DUP_LO(14) CFAR_DUP_LO(13) CFAR_DUP_LO(12) CFAR_DUP_LO(11) CFAR_DUP_LO(10) CFAR_DUP_LO(9)
        CFAR_DUP_LO(8) CFAR_DUP_LO(7) CFAR_DUP_LO(6) CFAR_DUP_LO(5) CFAR_DUP_LO(4) CFAR_DUP_LO(3) CFAR_DUP_LO(2) CFAR_DUP_LO(1)
        default: break;
      }
#undef CFAR_DUP_LO
    }
    if (last_col_thread >= NT - 1) {  // (uniform; a row that nearly fills the workgroup) what no thread is there to write, right of the last thread's column: the row total
      const int c0 = padl + NT - 1, ncol = RS - 1 - c0;  // whole columns c0 + 1 .., and the rows above TB of column c0 itself (thread NT's first values)
      const uint32_t tot_s = (uint32_t)tot << sh;
      for (int k = tid; k < nrows * ncol + P.rmax; k += NT) {
        if (k < nrows * ncol) lp[(k / ncol) * RS + c0 + 1 + k % ncol] = tot_s;
        else lp0[(TB + k - nrows * ncol) * RS + c0] = tot_s;
      }
    }
#pragma unroll
    for (int e = 0; e < TB; e++) asm volatile("" : "+v"(nq[e]));  // (kept in registers across the barrier: recomputing them from the bytes costs two instructions a bin)
    __syncthreads();
    if (P.stop == 1) { prev_total = -1; return; }
    // ---- the integer test on every bin of the thread (bit e of h <-> bin TB t + e) ----
    // (groups of GB bins, the next group's sixteen-odd reads in flight while this group's arithmetic runs: left to itself the compiler waits for
    // every four reads - fourteen LDS round trips a row with nothing else to do at two waves per SIMD)
    uint32_t h = 0;
    {
      constexpr int GB = 4, NG = TB / GB;  // (two groups = sixteen two-row reads: what the LDS counter can have outstanding)
      uint32_t va[2][GB], vb[2][GB], vc[2][GB], vd[2][GB];
#pragma unroll
      for (int k = 0; k < GB; k++) {
        const int e = TB - 1 - k;
        va[0][k] = pA[e * RS]; vb[0][k] = pB[e * RS]; vc[0][k] = pC[e * RS]; vd[0][k] = pD[e * RS];
      }
#pragma unroll
      for (int gi = 0; gi < NG; gi++) {
        const int cur = gi & 1, nxt = cur ^ 1;
        if (gi + 1 < NG) {
#pragma unroll
          for (int k = 0; k < GB; k++) {
            const int e = TB - 1 - ((gi + 1) * GB + k);
            va[nxt][k] = pA[e * RS]; vb[nxt][k] = pB[e * RS]; vc[nxt][k] = pC[e * RS]; vd[nxt][k] = pD[e * RS];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < GB; k++) {
          const int e = TB - 1 - (gi * GB + k);
          uint32_t r;  // bit 31 <=> (S << sh) <= I^2 kopen
          asm("v_add3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(nq[e]), "v"(vb[cur][k]), "v"(vd[cur][k]));
          r = r - va[cur][k] - vc[cur][k];
          h = __builtin_amdgcn_alignbit(h, r, 31);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    h &= vmask;
    if (P.stop == 2) { prev_total = (h ^ (uint32_t)nq[0]) == 0x12345u ? 0 : -1; return; }  // (a condition the compiler cannot decide: with `h == ~0u` - h has 28 bits - it moved the whole test behind this return)
    // ---- what it leaves, in bin order: one lane per candidate decides, the hits close ranks ----
    const int c = __popc(h);
    int total = 0;
    if (__builtin_amdgcn_ballot_w64(c != 0)) {
      const int inc = wave_inclusive_scan(c);
      const int ncand = __builtin_amdgcn_readlane(inc, 63);
      if (ncand <= 64) {
        {
          l_u32* out = lcand + wv * 64 + (inc - c);
          uint32_t hh = h;
          while (hh) {
            const int e = __ffs((int)hh) - 1;
            hh &= hh - 1u;
            *out++ = (uint32_t)(TB * tid + e);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        bool hit = false;
        int bin = 0, iv2 = 0;
        if (lane < ncand) { bin = (int)lcand[wv * 64 + lane]; hit = decide(bin, &iv2); }
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
        total = __popcll(hm);
        if (hit) {
          const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(hm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hm, 0u));
          // (the intensity back out of its square: the square is exact, the root of a perfect square <= 255^2 within an ulp of the integer)
          lstage[wv * CFAR_SEG_CAP + pos] = (uint32_t)bin | ((uint32_t)(__builtin_sqrtf((float)iv2) + 0.5f) << 16);
        }
      } else {  // a row of clutter: every thread walks its own candidates; the hit masks go out as they are and the emit kernel walks them
        uint32_t hh = h;
        while (hh) {
          const int e = __ffs((int)hh) - 1;
          hh &= hh - 1u;
          int iv2;
          if (!decide(TB * tid + e, &iv2)) h &= ~(1u << e);
        }
        hmask[(size_t)grow * NT + tid] = h;
        total = CFAR_SEG_MASKS | __builtin_amdgcn_readlane(wave_inclusive_scan(__popc(h)), 63);  // the flag: the masks hold the segment
      }
    }
    prev_total = total; prev_row = grow;
  };
  while (true) {
    if (grow >= rows) break;
    trip(buf0); grow += gridDim.x;
    if (grow >= rows) break;
    trip(buf1); grow += gridDim.x;
    if (grow >= rows) break;
    trip(buf2); grow += gridDim.x;
  }
  if (prev_total >= 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const size_t sidx = (size_t)prev_row * NW + wv;
    if (lane < prev_total && !(prev_total & CFAR_SEG_MASKS)) recs[sidx * CFAR_SEG_CAP + lane] = lstage[wv * CFAR_SEG_CAP + lane];
    if (lane == 0) seg_count[sidx] = prev_total;
  }
}

// segments -> points: lane <-> segment (64 consecutive segments per wave); a lane walks its segment's records. Segments that overflowed their
// slot are taken by the whole wave afterwards, from the threads' hit masks and the image.
__global__ __launch_bounds__(CFAR_BLOCK) void cfar_emit_recs_kernel(const uint8_t* __restrict__ polar, CfarParams P, const double* __restrict__ trig,
                                                                    const int* __restrict__ seg_count, const int* __restrict__ seg_base,
                                                                    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ hmask,
                                                                    float* __restrict__ xyi, int cap, long long n_segs, int NW, int TB) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long g0 = ((long long)blockIdx.x * (CFAR_BLOCK / 64) + wv) * 64;
  if (g0 >= n_segs) return;
  const long long gs = g0 + lane;
  const bool valid = gs < n_segs;
  const int craw = valid ? seg_count[gs] : 0, base = valid ? seg_base[gs] : 0;
  const int c = craw & (CFAR_SEG_MASKS - 1);
  if (c > 0 && !(craw & CFAR_SEG_MASKS)) {
    const long long row = gs / NW;
    const int img = (int)(row / P.A), az = (int)(row - (long long)img * P.A);
    const double cos_t = trig[2 * az], sin_t = trig[2 * az + 1];  // theta = (az + 1) / A * 2 pi, host libm (cfar.cpp:40)
    float* out = xyi + 3 * (size_t)img * cap;  // image i writes at most `cap` points at xyi + i * cap * 3
    const uint32_t* rr = recs + (size_t)gs * CFAR_SEG_CAP;
    auto put = [&](uint32_t rec, int k) __attribute__((always_inline)) {
      const int o = base + k;
      if (o < cap) {
        const double range = P.range_res * (double)(rec & 0xFFFFu);
        out[3 * (size_t)o + 0] = (float)(range * cos_t);  // cfar.cpp:63-65
        out[3 * (size_t)o + 1] = (float)(range * sin_t);
        out[3 * (size_t)o + 2] = (float)(rec >> 16);
      }
    };
    // the first eight slots of the segment in one round trip (a segment holds four or five records on the reference's preset; the slots exist whether
    // written or not, and a slot is 8-byte aligned for every shape), the rest one by one
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2* r2 = reinterpret_cast<const u32x2*>(rr);
    const u32x2 p0 = r2[0], p1 = r2[1], p2 = r2[2], p3 = r2[3];
    const uint32_t first[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
#pragma unroll
    for (int k = 0; k < 8; k++) if (k < c) put(first[k], k);
    for (int k = 8; k < c; k++) put(rr[k], k);
  }
  unsigned long long ov = __builtin_amdgcn_ballot_w64((craw & CFAR_SEG_MASKS) != 0);
  while (ov) {
    const int l = __ffsll((long long)ov) - 1;
    ov &= ov - 1ull;
    const long long sg = g0 + l;
    const long long row = sg / NW;
    const int swv = (int)(sg - row * NW), sbase = __builtin_amdgcn_readlane(base, l);
    const int img = (int)(row / P.A), az = (int)(row - (long long)img * P.A);
    const double cos_t = trig[2 * az], sin_t = trig[2 * az + 1];
    float* out = xyi + 3 * (size_t)img * cap;
    const uint8_t* rowp = polar + row * P.R;
    uint32_t h = hmask[(size_t)row * (64 * NW) + 64 * swv + lane];
    const int hc = __popc(h);
    int o = wave_inclusive_scan(hc) - hc + sbase;
    while (h) {
      const int e = __ffs((int)h) - 1;
      h &= h - 1u;
      const int i = TB * (64 * swv + lane) + e;
      if (o < cap) {
        const double range = P.range_res * (double)i;
        out[3 * (size_t)o + 0] = (float)(range * cos_t);
        out[3 * (size_t)o + 1] = (float)(range * sin_t);
        out[3 * (size_t)o + 2] = (float)rowp[i];
      }
      o++;
    }
  }
}

// a wave per four consecutive rows: the rows' masks -> their points, in range-bin order, at each row's offset of its image's cloud. (One
// single-wave workgroup per row - 614 400 of them per 1536 sweeps, nine detections each - spent its time being dispatched: 430 us; the
// counts, bases and mask words of a wave's four rows are in flight together here.)
constexpr int CFAR_EMIT_ROWS = 4;
__device__ __forceinline__ int cfar_emit_word(uint32_t m, int k, int o, int cap, const uint8_t* __restrict__ row, double range_res, double cos_t, double sin_t,
                                              float* __restrict__ xyi) {
  while (m) {
    const int b = __ffs((int)m) - 1;
    m &= m - 1;
    const int i = 32 * k + b;
    if (o < cap) {
      const double range = range_res * (double)i;
      xyi[3 * (size_t)o + 0] = (float)(range * cos_t);  // cfar.cpp:63-65
      xyi[3 * (size_t)o + 1] = (float)(range * sin_t);
      xyi[3 * (size_t)o + 2] = (float)row[i];
    }
    o++;
  }
  return o;
}
__global__ __launch_bounds__(CFAR_BLOCK) void cfar_emit_kernel(const uint8_t* __restrict__ polar, CfarParams P, const double* __restrict__ trig,
                                                               const int* __restrict__ row_count, const int* __restrict__ row_base,
                                                               const uint32_t* __restrict__ mask, float* __restrict__ xyi, int cap, long long rows) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long g0 = ((long long)blockIdx.x * (CFAR_BLOCK / 64) + wv) * CFAR_EMIT_ROWS;
  if (g0 >= rows) return;
  int cnt[CFAR_EMIT_ROWS], base[CFAR_EMIT_ROWS], any = 0;
#pragma unroll
  for (int r = 0; r < CFAR_EMIT_ROWS; r++) {
    const bool valid = g0 + r < rows;
    cnt[r] = valid ? row_count[g0 + r] : 0;
    base[r] = valid ? row_base[g0 + r] : 0;
    any |= cnt[r];
  }
  if (!any) return;
  const int wpl = (P.mask_words + 63) >> 6;  // consecutive mask words per lane
  const int w0 = min(P.mask_words, lane * wpl), w1 = min(P.mask_words, w0 + wpl);
  if (wpl <= 2) {  // (rows up to 4096 bins)
    uint32_t m[CFAR_EMIT_ROWS][2];
#pragma unroll
    for (int r = 0; r < CFAR_EMIT_ROWS; r++) {
      const uint32_t* mrow = mask + (size_t)(g0 + r) * P.mask_words;
      m[r][0] = (cnt[r] && w0 < w1) ? mrow[w0] : 0u;
      m[r][1] = (cnt[r] && w0 + 1 < w1) ? mrow[w0 + 1] : 0u;
    }
#pragma unroll
    for (int r = 0; r < CFAR_EMIT_ROWS; r++) {
      if (!cnt[r]) continue;  // (wave-uniform)
      const long long grow = g0 + r;
      const int img = (int)(grow / P.A), az = (int)(grow - (long long)img * P.A);
      const int c = __popc(m[r][0]) + __popc(m[r][1]);
      int o = wave_inclusive_scan(c) - c + base[r];
      if (c) {
        const double cos_t = trig[2 * az], sin_t = trig[2 * az + 1];  // theta = (az + 1) / A * 2 pi, host libm (cfar.cpp:40)
        const uint8_t* row = polar + grow * P.R;
        float* out = xyi + 3 * (size_t)img * cap;  // image i writes at most `cap` points at xyi + i * cap * 3
        o = cfar_emit_word(m[r][0], w0, o, cap, row, P.range_res, cos_t, sin_t, out);
        cfar_emit_word(m[r][1], w0 + 1, o, cap, row, P.range_res, cos_t, sin_t, out);
      }
    }
    return;
  }
  for (int r = 0; r < CFAR_EMIT_ROWS; r++) {
    if (!cnt[r]) continue;
    const long long grow = g0 + r;
    const int img = (int)(grow / P.A), az = (int)(grow - (long long)img * P.A);
    const uint32_t* mrow = mask + (size_t)grow * P.mask_words;
    int c = 0;
    for (int k = w0; k < w1; k++) c += __popc(mrow[k]);
    int o = wave_inclusive_scan(c) - c + base[r];
    if (c == 0) continue;
    const double cos_t = trig[2 * az], sin_t = trig[2 * az + 1];
    const uint8_t* row = polar + grow * P.R;
    float* out = xyi + 3 * (size_t)img * cap;
    for (int k = w0; k < w1; k++) o = cfar_emit_word(mrow[k], k, o, cap, row, P.range_res, cos_t, sin_t, out);
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
  for (int i = i0; i < i1; i++) s += row_count[i] & 0x3FFFFFFF;  // (bit 30: CFAR_SEG_MASKS of the owner-layout detector's segments)
  int tot;
  int o = block_exclusive_scan(s, red_i, &tot);
  for (int i = i0; i < i1; i++) { row_base[i] = o; o += row_count[i] & 0x3FFFFFFF; }
  if (threadIdx.x == 0) *d_total = tot;
}

// kernel parameters from the context's settings: the gates of cfar.cpp:45 as integer bounds (evaluated here with the reference's own
// double expressions - both are monotone in the bin / the intensity), the CA scaling factor (cfar.cpp:12-16, :32) by host libm
int cfar_params(cfear_ctx* ctx, int window_size, int nb_guard_cells, float false_alarm_rate, double max_distance, CfarParams* out) {
  if (window_size < 1 || nb_guard_cells < 0 || !(false_alarm_rate > 0.f))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: window_size >= 1, nb_guard_cells >= 0, false_alarm_rate > 0 required");
  if (ctx->R > CFAR_MAX_R) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar: more than 16384 range bins");
  if (window_size > (1 << 20) || nb_guard_cells > (1 << 20)) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar: window_size / nb_guard_cells beyond 2^20");
  CfarParams P;
  P.A = ctx->A; P.R = ctx->R; P.window = window_size; P.guard = nb_guard_cells;
  // float members of radarDriver::Parameters bound to const double& (radar_driver.cpp:54)
  P.range_res = (double)ctx->par.range_res;
  const double static_threshold = (double)ctx->par.z_min, min_distance = (double)ctx->par.min_distance;
  const double N = (double)(window_size * 2);  // CFARFilter::getCAScalingFactor
  P.scaling = N * (pow((double)false_alarm_rate, -1. / N) - 1.);
  P.kf = (float)(P.scaling * 0.5 / (double)window_size);
  P.ilo = P.R; P.ihi = -1;
  for (int i = 0; i < P.R; i++) {
    const double range = P.range_res * (double)i;
    if (range > min_distance && range < max_distance) { if (P.ilo == P.R) P.ilo = i; P.ihi = i; }
  }
  P.iv_min = 256;
  for (int v = 255; v >= 0; v--) if ((double)v > static_threshold) P.iv_min = v;
  P.mask_words = 2 * ((P.R + 63) / 64);
  P.ipt = (P.R + CFAR_BLOCK - 1) / CFAR_BLOCK;
  P.ipt |= 1;
  // owner-layout detector: the scale of the integer test (sh < 0: not usable - the old kernels run)
  P.sh = -1; P.kopen = 0;
  { static const int stop = getenv("CFEAR_CFAR_STOP") ? atoi(getenv("CFEAR_CFAR_STOP")) : 0; P.stop = stop; }
  {
    const double smax = 2.0 * (double)window_size * 65025.0, kf = P.scaling * 0.5 / (double)window_size;
    for (int sh = 12; sh >= 0 && P.sh < 0; sh--) {
      if (smax * (double)(1 << sh) >= 2147483648.0) continue;
      const double k = ceil((1.0 + 4e-6) * (double)(1 << sh) / kf);
      if (k >= 64.0 && k <= 33025.0) { P.sh = sh; P.kopen = (int)k; }  // I^2 kopen < 2^31; at least six bits of the threshold
    }
  }
  *out = P;
  return CFEAR_OK;
}

// ---- which instantiation of the owner-layout detector a row length takes (threads x bins per thread just above R) ----
struct CfarOwnerShape { int TB, NW; };

const CfarOwnerShape* cfar_owner_shape(int R, int reach /* guard + window, or 0: the shape with the largest scratch for this row length */) {
  static const CfarOwnerShape shapes[4] = {{28, 2}, {20, 3}, {16, 4}, {32, 4}};
  static const int forced = getenv("CFEAR_CFAR_SHAPE") ? atoi(getenv("CFEAR_CFAR_SHAPE")) : 0;  // bins per thread (tools/gpu_time_cfar.py: A/B of the shapes)
  // Measured at 3360 bins (profiles/r06_cfar_shapes.txt): two waves x 28 bins per thread is the fastest shape while the windows are short; with long
  // windows (the reference's sweep goes to 500 bins) the bins a row end clips are hundreds, the integer test is loose for them, and a wave of 1792
  // bins collects more than the 64 candidates one pass decides - four waves x 16 bins then.
  const CfarOwnerShape* best = nullptr;
  for (const CfarOwnerShape& s : shapes) {
    if (R >= 64 * s.NW * s.TB) continue;
    if (forced ? s.TB == forced : false) return &s;
    if (reach > 128 && s.NW < 4) continue;
    if (reach == 0 ? (!best || s.NW > best->NW) : (!best || s.NW * s.TB < best->NW * best->TB)) best = &s;
  }
  return best;
}
size_t cfar_owner_ints_per_row(const CfarOwnerShape& s) { return (size_t)s.NW * (2 + CFAR_SEG_CAP) + 64 * (size_t)s.NW; }
int cfar_device_cus(int device) {
  static int cus[64] = {0};
  if (device < 0 || device >= 64) return 256;
  if (!cus[device]) { int n = 0; cus[device] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n > 0) ? n : 256; }
  return cus[device];
}
size_t cfar_owner_lds_bytes(int nrows, int NW, int RS) { return sizeof(uint32_t) * ((size_t)nrows * RS + 64 * NW + (size_t)NW * CFAR_SEG_CAP + 16); }
// (row, column) offsets of the four window bounds for TB bins per thread: r in (-TB, TB), the combination with the fewest rows
void cfar_owner_offsets(CfarParams* P, int TB) {
  const int off[4] = {-P->guard - P->window, -P->guard, P->guard, P->guard + P->window};
  int best = 1 << 30;
  for (int m = 0; m < 16; m++) {
    int r[4], lo = 0, hi = 0;
    for (int c = 0; c < 4; c++) {
      r[c] = ((off[c] % TB) + TB) % TB;
      if (((m >> c) & 1) && r[c] > 0) r[c] -= TB;
      lo = r[c] < lo ? r[c] : lo; hi = r[c] > hi ? r[c] : hi;
    }
    if (hi - lo < best) {
      best = hi - lo; P->rmin = lo; P->rmax = hi;
      for (int c = 0; c < 4; c++) { P->cr[c] = r[c]; P->cq[c] = (off[c] - r[c]) / TB; }
    }
  }
}
template <int TB, int NW, int RS, bool FULL>
int cfar_launch_owner_tf(cfear_ctx* ctx, const CfarParams& P0, int padl, const uint8_t* d_polar, size_t rows, int* seg_count, uint32_t* recs, uint32_t* hmask, hipStream_t stream) {
  constexpr int NT = 64 * NW;
  CfarParams P = P0;
  cfar_owner_offsets(&P, TB);
  const size_t lds = cfar_owner_lds_bytes(TB + P.rmax - P.rmin, NW, RS);
  if (lds > 64 * 1024)
    CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cfar_detect_owner_kernel<TB, NW, RS, FULL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int per_cu = (int)((160 * 1024) / (lds + 512));  // persistent workgroups: what a compute unit's LDS holds, up to 32 waves
  if (per_cu * NW > 32) per_cu = 32 / NW;
  if (per_cu < 1) per_cu = 1;
  { static const int force = getenv("CFEAR_CFAR_PER_CU") ? atoi(getenv("CFEAR_CFAR_PER_CU")) : 0; if (force > 0 && force < per_cu) per_cu = force; }  // (profiling: fewer resident workgroups)
  size_t grid = (size_t)cfar_device_cus(ctx->device) * per_cu;
  if (grid > rows) grid = rows;
  hipLaunchKernelGGL((cfar_detect_owner_kernel<TB, NW, RS, FULL>), dim3((unsigned)grid), dim3(NT), lds, stream, d_polar, P, (int)rows, padl, seg_count, recs, hmask);
  return CFEAR_OK;
}
template <int TB, int NW, int RS>
int cfar_launch_owner_t(cfear_ctx* ctx, const CfarParams& P, int padl, const uint8_t* d_polar, size_t rows, int* seg_count, uint32_t* recs, uint32_t* hmask, hipStream_t stream) {
  return P.R % TB == 0 ? cfar_launch_owner_tf<TB, NW, RS, true>(ctx, P, padl, d_polar, rows, seg_count, recs, hmask, stream)
                       : cfar_launch_owner_tf<TB, NW, RS, false>(ctx, P, padl, d_polar, rows, seg_count, recs, hmask, stream);
}
// columns of the transposed prefix array this configuration needs: one per thread that holds bins (and the one that holds bin R), and on either
// side the window's reach in threads + 1; rounded up to a multiple of 64
int cfar_owner_columns(const CfarParams& P, const CfarOwnerShape& s, int* padl) {
  const int pad = (P.guard + P.window + s.TB - 1) / s.TB + 1, data = P.R / s.TB + 1;
  const int rs = (data + 2 * pad + 63) / 64 * 64;
  *padl = (rs - data) / 2;
  return rs;
}
// true when the owner-layout detector takes this configuration (dword rows and base, a usable integer scale, an instantiation for its columns)
bool cfar_owner_usable(const CfarParams& P, const uint8_t* d_polar, const CfarOwnerShape** shape) {
  static const bool off = getenv("CFEAR_CFAR_OLD_KERNELS") != nullptr;  // tools/gpu_time_cfar.py: A/B timing against the round-5 kernels
  if (off || (P.R & 3) != 0 || (reinterpret_cast<uintptr_t>(d_polar) & 3) != 0 || P.sh < 0 || P.iv_min > 255) return false;
  const CfarOwnerShape* s = cfar_owner_shape(P.R, P.guard + P.window);
  if (!s) return false;
  int padl;
  if (cfar_owner_columns(P, *s, &padl) > 64 * s->NW + 128) return false;
  *shape = s;
  return true;
}
int cfar_launch_owner(cfear_ctx* ctx, const CfarParams& P, const CfarOwnerShape& s, const uint8_t* d_polar, size_t rows, int* seg_count, uint32_t* recs,
                      uint32_t* hmask, hipStream_t stream) {
  int padl;
  const int rs = cfar_owner_columns(P, s, &padl);
#define CFAR_OWNER_CASE(TB_, NW_) \
  if (s.TB == TB_ && s.NW == NW_) { \
    if (rs <= 64 * NW_) return cfar_launch_owner_t<TB_, NW_, 64 * NW_>(ctx, P, padl + (64 * NW_ - rs) / 2, d_polar, rows, seg_count, recs, hmask, stream); \
    if (rs <= 64 * NW_ + 64) return cfar_launch_owner_t<TB_, NW_, 64 * NW_ + 64>(ctx, P, padl + (64 * NW_ + 64 - rs) / 2, d_polar, rows, seg_count, recs, hmask, stream); \
    return cfar_launch_owner_t<TB_, NW_, 64 * NW_ + 128>(ctx, P, padl + (64 * NW_ + 128 - rs) / 2, d_polar, rows, seg_count, recs, hmask, stream); \
  }
  CFAR_OWNER_CASE(28, 2)
  CFAR_OWNER_CASE(20, 3)
  CFAR_OWNER_CASE(16, 4)
  CFAR_OWNER_CASE(32, 4)
#undef CFAR_OWNER_CASE
  return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar: no detector instantiation for this row length");
}
// detect -> segment scan -> emit with the owner-layout detector; d_rows: cfear_cfar_scratch_ints ints. d_counts[i] = detections of image i.
int cfar_launch_owner_path(cfear_ctx* ctx, const CfarParams& P, const CfarOwnerShape& s, const uint8_t* d_polar, int n_scans, float* d_xyi, int capacity,
                           int* d_counts, int* d_rows, hipStream_t stream, bool emit) {
  const size_t rows = (size_t)n_scans * P.A, segs = rows * s.NW;
  int* seg_count = d_rows; int* seg_base = seg_count + segs;
  uint32_t* recs = reinterpret_cast<uint32_t*>(seg_base + segs);
  uint32_t* hmask = recs + segs * CFAR_SEG_CAP;
  int rc = cfar_launch_owner(ctx, P, s, d_polar, rows, seg_count, recs, hmask, stream);
  if (rc != CFEAR_OK) return rc;
  hipLaunchKernelGGL(cfar_row_scan_kernel, dim3(n_scans), dim3(1024), 0, stream, seg_count, P.A * s.NW, seg_base, d_counts);
  if (emit)
    hipLaunchKernelGGL(cfar_emit_recs_kernel, dim3((unsigned)((segs + 255) / 256)), dim3(CFAR_BLOCK), 0, stream, d_polar, P, ctx->d_trig, seg_count, seg_base, recs, hmask,
                       d_xyi, capacity, (long long)segs, s.NW, s.TB);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}
size_t cfar_lds_bytes(const CfarParams& P) { return sizeof(uint32_t) * (size_t)(P.R + 1) + (size_t)((P.R + 3 + 7) & ~3); }
template <int TB>
int cfar_launch_fast(cfear_ctx* ctx, const CfarParams& P, const uint8_t* d_polar, size_t rows, int* d_count, uint32_t* d_mask, hipStream_t stream) {
  const int pad = (P.guard + P.window + 3) & ~3;
  const size_t lds = sizeof(uint32_t) * ((size_t)pad + CFAR_BLOCK * TB + pad + 4);  // the padded prefix array
  if (lds > 64 * 1024)
    CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cfar_detect_fast_kernel<TB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((cfar_detect_fast_kernel<TB>), dim3((unsigned)rows), dim3(CFAR_BLOCK), lds, stream, d_polar, P, pad, d_count, d_mask);
  return CFEAR_OK;
}
int cfar_launch_detect(cfear_ctx* ctx, const CfarParams& P, const uint8_t* d_polar, size_t rows, int* d_count, uint32_t* d_mask, hipStream_t stream) {
  // the lean kernel when the rows are dword-aligned, fit 256 threads x 16 (32) bins, and the window is not enormous (its pads live in LDS)
  static const bool general_only = getenv("CFEAR_CFAR_GENERAL_KERNEL") != nullptr;  // tools/gpu_time_cfar.py: A/B timing of the two detectors
  if ((P.R & 3) == 0 && (reinterpret_cast<uintptr_t>(d_polar) & 3) == 0 && P.guard + P.window <= 2048 && P.iv_min <= 255 && !general_only) {
    if (P.R <= CFAR_BLOCK * 16) return cfar_launch_fast<16>(ctx, P, d_polar, rows, d_count, d_mask, stream);
    if (P.R <= CFAR_BLOCK * 32) return cfar_launch_fast<32>(ctx, P, d_polar, rows, d_count, d_mask, stream);
  }
  const size_t lds = cfar_lds_bytes(P);
  if (lds > 64 * 1024)
    CFEAR_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(cfar_detect_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(cfar_detect_kernel, dim3((unsigned)rows), dim3(CFAR_BLOCK), lds, stream, d_polar, (long long)rows * P.R, P, d_count, d_mask);
  return CFEAR_OK;
}

int cfar_run(cfear_ctx* ctx, const uint8_t* d_polar, int window_size, int nb_guard_cells,
             float false_alarm_rate, double max_distance, cfear_cloud** out) {
  if (!out) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: null output");
  *out = nullptr;
  CfarParams P;
  int rc = cfar_params(ctx, window_size, nb_guard_cells, false_alarm_rate, max_distance, &P);
  if (rc != CFEAR_OK) return rc;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int* d_tmp = nullptr;  // the detector's row scratch + the total (a block of the context's pool)
  size_t tmp_bytes = 0;
  const size_t row_ints = cfear_cfar_scratch_ints(ctx, 1);
  { void* blk = nullptr; rc = cfear_pool_alloc(ctx, sizeof(int) * (row_ints + 1), &blk, &tmp_bytes); if (rc != CFEAR_OK) return rc; d_tmp = static_cast<int*>(blk); }
  int* d_total = d_tmp + row_ints;
  const CfarOwnerShape* shape = nullptr;
  const bool owner = cfar_owner_usable(P, d_polar, &shape);
  int* d_count = d_tmp; int* d_base = d_tmp + P.A;
  uint32_t* d_mask = reinterpret_cast<uint32_t*>(d_tmp + 2 * P.A);
  if (owner) {
    rc = cfar_launch_owner_path(ctx, P, *shape, d_polar, 1, nullptr, 0, d_total, d_tmp, ctx->stream, false);
  } else {
    rc = cfar_launch_detect(ctx, P, d_polar, (size_t)P.A, d_count, d_mask, ctx->stream);
    if (rc == CFEAR_OK) hipLaunchKernelGGL(cfar_row_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_count, P.A, d_base, d_total);
  }
  if (rc != CFEAR_OK) { cfear_pool_free(ctx, d_tmp, tmp_bytes); return rc; }
  int total = 0;
  hipError_t e = hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { cfear_pool_free(ctx, d_tmp, tmp_bytes); return cfear_fail(ctx, CFEAR_ERR_HIP, "filter_cfar detect pass", e); }
  cfear_cloud* c = nullptr;
  rc = cfear_cloud_alloc(ctx, total, &c);
  if (rc != CFEAR_OK) { cfear_pool_free(ctx, d_tmp, tmp_bytes); return rc; }
  if (total > 0) {
    if (owner) {
      const size_t segs = (size_t)P.A * shape->NW;
      int* seg_count = d_tmp; int* seg_base = seg_count + segs;
      uint32_t* recs = reinterpret_cast<uint32_t*>(seg_base + segs);
      hipLaunchKernelGGL(cfar_emit_recs_kernel, dim3((unsigned)((segs + 255) / 256)), dim3(CFAR_BLOCK), 0, ctx->stream, d_polar, P, ctx->d_trig, seg_count, seg_base, recs,
                         recs + segs * CFAR_SEG_CAP, c->d_xyi, c->cap, (long long)segs, shape->NW, shape->TB);
    } else {
      hipLaunchKernelGGL(cfar_emit_kernel, dim3((P.A + 4 * CFAR_EMIT_ROWS - 1) / (4 * CFAR_EMIT_ROWS)), dim3(CFAR_BLOCK), 0, ctx->stream, d_polar, P, ctx->d_trig, d_count, d_base, d_mask,
                         c->d_xyi, c->cap, (long long)P.A);
    }
  }
  e = hipMemcpyAsync(c->d_n, d_total, sizeof(int), hipMemcpyDeviceToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  cfear_pool_free(ctx, d_tmp, tmp_bytes);
  if (e != hipSuccess) { cfear_cloud_release(ctx, c); return cfear_fail(ctx, CFEAR_ERR_HIP, "filter_cfar emit pass", e); }
  *out = c;
  return CFEAR_OK;
}

}  // namespace

// ints of row scratch the batched filter needs for n_scans images: row counts, row bases, hit masks
__attribute__((visibility("hidden"))) size_t cfear_cfar_scratch_ints(const cfear_ctx* ctx, size_t n_scans) {
  size_t per_row = 2 + 2 * (size_t)((ctx->R + 63) / 64);  // mask-based kernels
  const CfarOwnerShape* s = cfar_owner_shape(ctx->R, 0);  // owner-layout detector: segment counts, bases, record slots, overflow masks (its widest shape)
  if (s && cfar_owner_ints_per_row(*s) > per_row) per_row = cfar_owner_ints_per_row(*s);
  return n_scans * (size_t)ctx->A * per_row;
}
// the batched filter on `stream` with the caller's row scratch (cfear_cfar_scratch_ints): detect pass (the only one that reads the
// images), one row scan per image, emit pass (cfear_filter_cfar_batch_device; the CA-CFAR stage of the batched odometry objects)
__attribute__((visibility("hidden"))) int cfear_launch_cfar_batch(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, int window_size, int nb_guard_cells,
                                                                  float false_alarm_rate, double max_distance, float* d_xyi, int capacity, int* d_counts,
                                                                  int* d_rows, hipStream_t stream) {
  if ((reinterpret_cast<uintptr_t>(d_polar) & 3) != 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar_batch: polar buffer must be 4-byte aligned");
  if ((long long)n_scans * ctx->A > 0x7FFFFFFFLL) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar_batch: too many rows");
  CfarParams P;
  int rc = cfar_params(ctx, window_size, nb_guard_cells, false_alarm_rate, max_distance, &P);
  if (rc != CFEAR_OK) return rc;
  const size_t rows = (size_t)n_scans * ctx->A;
  {
    const CfarOwnerShape* shape = nullptr;
    if (cfar_owner_usable(P, d_polar, &shape))
      return cfar_launch_owner_path(ctx, P, *shape, d_polar, n_scans, d_xyi, capacity, d_counts, d_rows, stream, true);
  }
  int* d_count = d_rows; int* d_base = d_count + rows;
  uint32_t* d_mask = reinterpret_cast<uint32_t*>(d_base + rows);
  rc = cfar_launch_detect(ctx, P, d_polar, rows, d_count, d_mask, stream);
  if (rc != CFEAR_OK) return rc;
  hipLaunchKernelGGL(cfar_row_scan_kernel, dim3(n_scans), dim3(1024), 0, stream, d_count, P.A, d_base, d_counts);
  hipLaunchKernelGGL(cfar_emit_kernel, dim3((unsigned)((rows + 4 * CFAR_EMIT_ROWS - 1) / (4 * CFAR_EMIT_ROWS))), dim3(CFAR_BLOCK), 0, stream, d_polar, P, ctx->d_trig, d_count, d_base,
                     d_mask, d_xyi, capacity, (long long)rows);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

extern "C" {

int cfear_filter_cfar_device(cfear_ctx* ctx, const uint8_t* d_polar, int window_size, int nb_guard_cells, float false_alarm_rate,
                             double max_distance, cfear_cloud** cloud) {
  if (!ctx || !d_polar) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: bad argument");
  if ((reinterpret_cast<uintptr_t>(d_polar) & 3) != 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar: polar buffer must be 4-byte aligned");
  return cfar_run(ctx, d_polar, window_size, nb_guard_cells, false_alarm_rate, max_distance, cloud);
}

int cfear_filter_cfar_batch_device(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, int window_size, int nb_guard_cells,
                                   float false_alarm_rate, double max_distance, float* d_xyi, int capacity, int* d_counts) {
  if (!ctx || !d_polar || !d_xyi || !d_counts || n_scans <= 0 || capacity <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_cfar_batch: bad argument");
  if ((long long)n_scans * ctx->A > 0x7FFFFFFFLL) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "filter_cfar_batch: too many rows");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t need = cfear_cfar_scratch_ints(ctx, (size_t)n_scans);
  if (need > ctx->cfar_rows_cap) {  // row counts, row bases and hit masks of the whole batch
    if (ctx->d_cfar_rows) { CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->d_cfar_rows); }
    ctx->d_cfar_rows = nullptr; ctx->cfar_rows_cap = 0;
    if (hipMalloc(&ctx->d_cfar_rows, sizeof(int) * need) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc cfar rows");
    ctx->cfar_rows_cap = need;
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
  rc = cfear_upload_image(ctx, ctx->d_polar, h_polar, (size_t)ctx->A * ctx->R);
  if (rc != CFEAR_OK) return rc;
  return cfar_run(ctx, ctx->d_polar, window_size, nb_guard_cells, false_alarm_rate, max_distance, cloud);
}

}  // extern "C"
