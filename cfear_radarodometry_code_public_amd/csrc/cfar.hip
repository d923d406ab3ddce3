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
  for (int i = i0; i < i1; i++) s += row_count[i];
  int tot;
  int o = block_exclusive_scan(s, red_i, &tot);
  for (int i = i0; i < i1; i++) { row_base[i] = o; o += row_count[i]; }
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
  *out = P;
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
  int* d_tmp = nullptr;  // row counts, row bases, total, masks (a block of the context's pool)
  size_t tmp_bytes = 0;
  { void* blk = nullptr; rc = cfear_pool_alloc(ctx, sizeof(int) * ((2 + (size_t)P.mask_words) * P.A + 1), &blk, &tmp_bytes); if (rc != CFEAR_OK) return rc; d_tmp = static_cast<int*>(blk); }
  int* d_count = d_tmp; int* d_base = d_tmp + P.A; int* d_total = d_tmp + 2 * P.A;
  uint32_t* d_mask = reinterpret_cast<uint32_t*>(d_total + 1);
  rc = cfar_launch_detect(ctx, P, d_polar, (size_t)P.A, d_count, d_mask, ctx->stream);
  if (rc != CFEAR_OK) { cfear_pool_free(ctx, d_tmp, tmp_bytes); return rc; }
  hipLaunchKernelGGL(cfar_row_scan_kernel, dim3(1), dim3(1024), 0, ctx->stream, d_count, P.A, d_base, d_total);
  int total = 0;
  hipError_t e = hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { cfear_pool_free(ctx, d_tmp, tmp_bytes); return cfear_fail(ctx, CFEAR_ERR_HIP, "filter_cfar detect pass", e); }
  cfear_cloud* c = nullptr;
  rc = cfear_cloud_alloc(ctx, total, &c);
  if (rc != CFEAR_OK) { cfear_pool_free(ctx, d_tmp, tmp_bytes); return rc; }
  if (total > 0)
    hipLaunchKernelGGL(cfar_emit_kernel, dim3((P.A + 4 * CFAR_EMIT_ROWS - 1) / (4 * CFAR_EMIT_ROWS)), dim3(CFAR_BLOCK), 0, ctx->stream, d_polar, P, ctx->d_trig, d_count, d_base, d_mask,
                       c->d_xyi, c->cap, (long long)P.A);
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
  return n_scans * (size_t)ctx->A * (2 + 2 * (size_t)((ctx->R + 63) / 64));
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
