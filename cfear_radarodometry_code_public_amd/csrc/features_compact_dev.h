// features_compact_dev.h -- MapPointNormal::ComputeNormals + ComputeSearchTreeFromCells (pointnormal.cpp:265-297, :7-63,
// :151-162) for clouds of up to CFEAR_CPT_CAP points with a working set of < 80 KB of LDS, so that TWO feature workgroups
// share a compute unit (the path is a chain of short phases separated by barriers: a second workgroup works while the
// first one waits). Same results as features_block (features_dev.h), which stays as the path for bigger clouds.
//
// What makes it small:
//  * the dense voxel grid is a BITMAP (one bit per voxel, <= 32768 voxels = 4 KB) plus a popcount prefix per word: the rank
//    of a voxel among the occupied ones is one word, one prefix and a popcount. The counting sort runs over the occupied
//    voxels only (16-bit counters), and the first occupied voxel >= k of a voxel row - what the radius search needs - is
//    the same rank query instead of a binary search in a voxel list;
//  * sorted points are float2 + one intensity byte (9 bytes instead of 12), every index array is 16-bit;
//  * a sample's centroid and candidate row ranges are recomputed by whoever needs them (a few LDS reads) instead of being
//    stored per sample.
#pragma once
#include <type_traits>

#include "features_dev.h"

namespace cfear_dev {

#define CFEAR_CPT_CAP 4864      // points (the reference configuration has at most A * k = 4800)
#define CFEAR_CPT_VOXELS 32768  // voxels of the dense grid the bitmap covers

struct FeatLdsC {  // byte offsets into the LDS segment
  static constexpr size_t red_i = 0;                                        // 64 ints
  static constexpr size_t red_f = red_i + 64 * sizeof(int);                 // 64 floats
  static constexpr size_t bm = red_f + 64 * sizeof(float);                  // bitmap words (+1: rank(G) looks one past)
  static constexpr size_t bmp = bm + (CFEAR_CPT_VOXELS / 32 + 4) * 4;       // u16 prefix per bitmap word (+1)
  static constexpr size_t vst = bmp + (CFEAR_CPT_VOXELS / 32 + 8) * 2;      // u16 [cap + 2]: voxel starts (vst[c] .. vst[c + 1] = the slots of occupied voxel c)
  static constexpr size_t ord = vst + (CFEAR_CPT_CAP + 8) * 2;              // u16 [cap + 2]: parked voxel ids, unordered slots, candidate totals, chunk starts
  static constexpr size_t chk = ord + (CFEAR_CPT_CAP + 8) * 2;              // u16 [cap]: sample of every chunk; later the float cell means
  static constexpr size_t pw = chk + CFEAR_CPT_CAP * 2;                     // u8 [cap] moment weights (from the intensities) in sorted order
  static constexpr size_t pxy = (pw + CFEAR_CPT_CAP + 15) / 16 * 16;        // float2 [cap] points in sorted order; before that the bearing table of the cloud pass and the counting-sort counters, afterwards the grid counters
  static constexpr size_t total = pxy + CFEAR_CPT_CAP * 8;
};
static_assert(FeatLdsC::total <= 80384, "two feature workgroups per compute unit need <= 80,384 B each (LDS comes in 1,280-byte granules)");

// Returns false (before touching anything but registers) when the cloud does not fit this path: more than CFEAR_CPT_CAP
// points or a voxel grid beyond the bitmap. zeroed: the caller cleared bm already (saves a barrier).
__device__ __forceinline__ bool features_block_c(ScanDev* __restrict__ S, int n, const FeatureParams& P, const FeatureScratch& W,
                                                 unsigned char* lds, PhaseTimer* pt, const float* bounds, bool zeroed,
                                                 const PointRegs& PR) {
  typedef __attribute__((address_space(1))) double g_f64;
  typedef __attribute__((address_space(1))) float g_f32;
  typedef __attribute__((address_space(3))) unsigned l_u32;
  typedef __attribute__((address_space(3))) unsigned short l_u16;
  typedef __attribute__((address_space(3))) unsigned char l_u8;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) f32x2 l_f32x2;
  l_u32* const bm = (l_u32*)(lds + FeatLdsC::bm);
  l_u16* const bmp = (l_u16*)(lds + FeatLdsC::bmp);
  l_u16* const vst = (l_u16*)(lds + FeatLdsC::vst);
  l_u16* const ord = (l_u16*)(lds + FeatLdsC::ord);
  l_u16* const chk = (l_u16*)(lds + FeatLdsC::chk);
  l_u8* const pw = (l_u8*)(lds + FeatLdsC::pw);
  l_f32x2* const pxy = (l_f32x2*)(lds + FeatLdsC::pxy);
  g_f64* const g_part = (g_f64*)W.part;
  typedef __attribute__((address_space(1))) f32x2 g_f32x2;
  g_f32x2* const g_cen = (g_f32x2*)W.samples;  // voxel centroids (8-byte aligned: the scratch arrays start on 256 B)
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) u32x4 g_u32x4;
  g_u32x4* const g_rng = (g_u32x4*)W.rng;  // candidate ranges of a sample's window rows (16 bytes per sample)
  const int tid = threadIdx.x, nt = CFEAR_FEAT_BLOCK;
  const int ccap = min(CFEAR_CPT_CAP, W.cap);  // points / chunk records the arrays (LDS and the global partial sums) hold
  if (n <= 0 || n > ccap || PR.rounds <= 0) return false;  // block-uniform
  // ---- PCL VoxelGrid (pointnormal.cpp:277-280), leaf = radius_/downsample_factor ----
  const float leaf = (float)((double)P.radius / P.downsample_factor);
  const float inv = 1.0f / leaf;
  float mnx = 3.4e38f, mxx = -3.4e38f, mny = 3.4e38f, mxy = -3.4e38f;
  if (bounds) {  // block-uniform: the caller already knows the bounding box
    mnx = bounds[0]; mxx = bounds[1]; mny = bounds[2]; mxy = bounds[3];
  } else {
#pragma unroll
    for (int r = 0; r < CFEAR_PT; r++) {
      const float x = PR.x[r], y = PR.y[r];
      if (preg_on(PR, r)) { mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y); }
    }
    float bb[4] = {mnx, mxx, mny, mxy};
    block_bounds<CFEAR_FEAT_BLOCK>(bb, W.red_f);
    mnx = bb[0]; mxx = bb[1]; mny = bb[2]; mxy = bb[3];
  }
  const int min_b0 = (int)floorf(mnx * inv), max_b0 = (int)floorf(mxx * inv);
  const int min_b1 = (int)floorf(mny * inv), max_b1 = (int)floorf(mxy * inv);
  const int div0 = max_b0 - min_b0 + 1, div1 = max_b1 - min_b1 + 1;
  const long long Gll = (long long)div0 * (long long)div1;
  if (Gll > CFEAR_CPT_VOXELS || Gll <= 0) return false;  // block-uniform
  const int G = (int)Gll, NW = (G >> 5) + 1;  // bitmap words incl. the one rank(G) looks at
  if (!zeroed) {
    for (int i = tid; i < CFEAR_CPT_VOXELS / 32 + 4; i += nt) bm[i] = 0u;
    __syncthreads();
  }
  // the points are in registers (PointRegs): wave w holds a contiguous run of the cloud and meets it in index order, round
  // after round, lane after lane. That order is what makes the scatter below stable.
  constexpr int PT = CFEAR_PT;
  const int wv = tid >> 6, nwv = nt >> 6;
  auto pidx = [&](int r) -> int { return preg_idx(PR, r); };
  auto pon = [&](int r) -> bool { return preg_on(PR, r); };
  // ---- occupied voxels: bitmap (a point's voxel index stays in its thread's registers) ----
  int pv[PT];
#pragma unroll
  for (int r = 0; r < PT; r++) {
    pv[r] = 0;
    if (pon(r)) {
      const int ijk0 = (int)(floorf(PR.x[r] * inv) - (float)min_b0);
      const int ijk1 = (int)(floorf(PR.y[r] * inv) - (float)min_b1);
      pv[r] = ijk0 + ijk1 * div0;
      __hip_atomic_fetch_or(&bm[pv[r] >> 5], 1u << (pv[r] & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  if (pt) pt->mark();
  CFEAR_STOP_AT(2, true);
  // ---- popcount prefix per bitmap word; nv = occupied voxels = sample points ----
  // Counting-sort counters: one set of 16-bit counters per wave (where the sorted points go later) when they fit, else one
  // set for all. With a set per wave the sort is STABLE without a fix-up: a voxel's slots are handed out wave after wave
  // (the prefix below), inside a wave round after round (the LDS executes a wave's atomics in program order) and inside one
  // atomic instruction lane after lane. That last order is how gfx950 resolves same-address lanes (tools/micro/
  // lds_atomic_order.hip) but nothing documents it, so the result is checked (a point's predecessor in its voxel must have a
  // smaller index) and a block that finds a violation - or whose counters do not fit - ranks its points by counting instead.
  l_u32* const cw = (l_u32*)(lds + FeatLdsC::pxy);
  l_u16* const cw16 = (l_u16*)(lds + FeatLdsC::pxy);
  int nv, VS, NS;
  {
    const int wpt = (NW + nt - 1) / nt;
    const int w0 = tid * wpt, w1 = min(NW, w0 + wpt);
    int cnt = 0;
    for (int w = w0; w < w1; w++) cnt += __popc(bm[w]);
    int ex = block_exclusive_scan_1b<CFEAR_FEAT_BLOCK>(cnt, W.red_i, 0, &nv);  // (one-barrier scans: a barrier separates each from the one before)
    for (int w = w0; w < w1; w++) { bmp[w] = (unsigned short)ex; ex += __popc(bm[w]); }
    VS = (nv + 2) & ~1;  // counters per set (even: a set starts on a word)
    NS = (nwv == 8 && 8 * VS <= (int)(CFEAR_CPT_CAP * 8 / 2)) ? 8 : 1;
    for (int i = tid; i < NS * VS / 2; i += nt) cw[i] = 0u;
  }
  __syncthreads();
  const int cbase = NS > 1 ? wv * VS : 0;
  // rank of voxel index k among the occupied voxels = number of occupied voxels below k (k in 0..G)
  auto rank = [&](int k) -> int { return (int)bmp[k >> 5] + __popc(bm[k >> 5] & ((1u << (k & 31)) - 1u)); };
  // ---- counting sort over the occupied voxels ([3P] std::sort on the voxel index, pinned as stable) ----
#pragma unroll
  for (int r = 0; r < PT; r++) {
    if (pon(r)) {
      pv[r] = rank(pv[r]);  // compact voxel index from here on
      const int c1 = cbase + pv[r];
      __hip_atomic_fetch_add(&cw[c1 >> 1], 1u << (16 * (c1 & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  {  // voxel starts (vst[c], vst[nv] = n); a counter becomes the first slot its wave may hand out in that voxel. Two voxels
     // (one 32-bit word of every set) at a time, the sets' words in registers.
    l_u32* const vstw = (l_u32*)(lds + FeatLdsC::vst);
    const int nwords = (nv + 1) >> 1;  // words that hold a voxel
    const int wpt = (nwords + nt - 1) / nt;
    const int w0 = tid * wpt, w1 = min(nwords, w0 + wpt);
    auto starts = [&](auto nsc) {
      constexpr int NSC = decltype(nsc)::value;
      const int VW = VS >> 1;  // words per set
      int cnt = 0;
      for (int w = w0; w < w1; w++) {
#pragma unroll
        for (int q = 0; q < NSC; q++) { const unsigned c = cw[q * VW + w]; cnt += (int)(c & 0xFFFFu) + (int)(c >> 16); }
      }
      int tot;
      int o = block_exclusive_scan_1b<CFEAR_FEAT_BLOCK>(cnt, W.red_i, 0, &tot);
      for (int w = w0; w < w1; w++) {
        unsigned c[NSC];
        int sum_lo = 0;
#pragma unroll
        for (int q = 0; q < NSC; q++) { c[q] = cw[q * VW + w]; sum_lo += (int)(c[q] & 0xFFFFu); }
        int olo = o, ohi = o + sum_lo;
        vstw[w] = (unsigned)olo | ((unsigned)ohi << 16);
#pragma unroll
        for (int q = 0; q < NSC; q++) {
          cw[q * VW + w] = (unsigned)olo | ((unsigned)ohi << 16);
          olo += (int)(c[q] & 0xFFFFu); ohi += (int)(c[q] >> 16);
        }
        o = ohi;
      }
    };
    if (NS == 8) starts(std::integral_constant<int, 8>{}); else starts(std::integral_constant<int, 1>{});
    __syncthreads();  // (the starts are complete before vst[nv] is set: for an odd nv it is the high half of the last word, holding n already)
    if (tid == 0) { vst[nv] = (unsigned short)n; S->n_samples = nv; S->n_points = n; S->status = 0; }
  }
  int pos[PT];
#pragma unroll
  for (int r = 0; r < PT; r++) {
    pos[r] = 0;
    if (pon(r)) {
      const int c1 = cbase + pv[r];
      const unsigned old = __hip_atomic_fetch_add(&cw[c1 >> 1], 1u << (16 * (c1 & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      pos[r] = (int)((old >> (16 * (c1 & 1))) & 0xFFFFu);
      ord[pos[r]] = (unsigned short)pidx(r);
    }
  }
  __syncthreads();
  bool ordered = false;
  if (NS > 1) {  // block-uniform
    int bad = 0;
#pragma unroll
    for (int r = 0; r < PT; r++)
      if (pon(r)) bad |= ((pos[r] > (int)vst[pv[r]]) & ((int)ord[max(pos[r] - 1, 0)] >= pidx(r))) ? 1 : 0;
    ordered = !__syncthreads_or(bad);
  }
  // the point goes straight to its place in the sorted arrays (x, y as floats, the intensity - an integer 0..255 from the
  // filter's slots - as a byte); the counters are dead from here on
#pragma unroll
  for (int r = 0; r < PT; r++) {
    const int i = pidx(r);
    if (pon(r)) {
      int fin = pos[r];
      if (!ordered) {  // final slot = voxel start + number of voxel members with a smaller point index (rank by counting)
        const int a = (int)vst[pv[r]], b = (int)vst[pv[r] + 1];
        int c = 0;
        for (int q = a; q < b; q += 4) {  // four members per trip: independent LDS loads
          int o4[4];
#pragma unroll
          for (int u = 0; u < 4; u++) o4[u] = (int)ord[min(q + u, b - 1)];
#pragma unroll
          for (int u = 0; u < 4; u++) c += ((q + u < b) & (o4[u] < i)) ? 1 : 0;
        }
        fin = a + c;
      }
      // (one 8-byte store from the two bit patterns: built as a float vector, the optimizer widens the read of PR.x[r] into a
      // vector load that spans x[r + 1], which keeps the whole register array in scratch memory)
      ((__attribute__((address_space(3))) unsigned long long*)pxy)[fin] =
          (unsigned long long)__float_as_uint(PR.x[r]) | ((unsigned long long)__float_as_uint(PR.y[r]) << 32);
      // what the moments need of the intensity: the weight max(I - 60, 0) (pointnormal.cpp:15), an integer 0..195, or 1
      const int iw = preg_w(PR, r);
      pw[fin] = (unsigned char)(P.weight_intensity ? (iw > 60 ? iw - 60 : 0) : 1);
    }
  }
  __syncthreads();
  if (pt) { pt->mark(); pt->mark(); pt->mark(); }
  CFEAR_STOP_AT(3, true);
  // ---- radius search + cell statistics per sample point (pointnormal.cpp:286-296, :7-63) ----
  // One pass over the candidates with moments shifted by the sample point c:
  //   mean = c + S1/S0,  cov = S2/S0 - (S1/S0)(S1/S0)^T   (== sum w_i (x_i-u)(x_i-u)^T with sum w_i = 1)
  const float r2 = (float)((double)P.radius * (double)P.radius);
  const float rq = P.radius * 1.0001f;
  // centroid of sample v: float sums in ascending (voxel, point) order, divided by float(count) ([3P] PCL CentroidPoint)
  auto centroid = [&](int v, float& cx, float& cy) {
    const int a = (int)vst[v], b = (int)vst[v + 1];
    float sx = 0.f, sy = 0.f;
    int q = a;
    for (; q + 4 <= b; q += 4) {  // whole groups of four: four loads in flight, added in point order, nothing to mask
      f32x2 p[4];
#pragma unroll
      for (int u = 0; u < 4; u++) p[u] = pxy[q + u];
#pragma unroll
      for (int u = 0; u < 4; u++) { sx += p[u].x; sy += p[u].y; }
    }
    {  // the last one to three points
      f32x2 p[3];
#pragma unroll
      for (int u = 0; u < 3; u++) p[u] = pxy[min(q + u, b - 1)];
#pragma unroll
      for (int u = 0; u < 3; u++) { const bool on = q + u < b; sx = on ? sx + p[u].x : sx; sy = on ? sy + p[u].y : sy; }
    }
    const float cnt = (float)(b - a);
    cx = sx / cnt; cy = sy / cnt;
  };
  struct Win { int gx0, gx1, gy0, gy1; };
  auto window = [&](float cx, float cy) -> Win {
    Win w;
    w.gx0 = (int)(floorf((cx - rq) * inv) - (float)min_b0); w.gx1 = (int)(floorf((cx + rq) * inv) - (float)min_b0);
    w.gy0 = (int)(floorf((cy - rq) * inv) - (float)min_b1); w.gy1 = (int)(floorf((cy + rq) * inv) - (float)min_b1);
    w.gx0 = max(w.gx0, 0); w.gy0 = max(w.gy0, 0); w.gx1 = min(w.gx1, div0 - 1); w.gy1 = min(w.gy1, div1 - 1);
    return w;
  };
  // candidates of voxel row gy of the window: voxels gx0..gx1 are contiguous in the sorted order
  auto row_range = [&](const Win& w, int gy, int& a, int& b) {
    const int k0 = w.gx0 + gy * div0, k1 = w.gx1 + gy * div0 + 1;
    a = (int)vst[rank(k0)]; b = (int)vst[rank(k1)];
  };
  // the ranges of window rows gy .. gy + 3 at once (rows past the window: empty): the eight rank queries and the eight
  // start lookups are independent LDS reads - row after row they were a chain of dependent ones
  auto row_ranges4 = [&](const Win& w, int gy, int (&a)[4], int (&b)[4]) {
    int r0[4], r1[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int g = min(gy + u, w.gy1);
      r0[u] = rank(w.gx0 + g * div0); r1[u] = rank(w.gx1 + g * div0 + 1);
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const bool on = (gy + u <= w.gy1) & (w.gx0 <= w.gx1);
      const int sa = (int)vst[r0[u]], sb = (int)vst[r1[u]];
      a[u] = sa; b[u] = on ? sb : sa;
    }
  };
  // Candidate counts range from 1 to ~1000 per sample point, so the work is cut into chunks of at most C candidates:
  // (1) per sample the candidate total, (2) a scan turns the totals into a chunk list, (3) one lane per chunk accumulates
  // partial moments, (4) the epilogue adds a sample's partials in chunk order (deterministic).
  // (chunks of 16: with 512 lanes and some 300-1500 samples of 6..300 candidates each, shorter chunks spread the work more
  // evenly over the lanes than chunks of 32 - fewer lanes wait for the longest chunk of their wave)
  int C = 16, CS = 4, NC, NA;  // chunk size (a power of two: 1 << CS, so that counting chunks is a shift, not an integer division), chunks, samples that have chunks ("active": only they can become cells)
  bool listed = true;  // the active samples are listed behind the chunks (when both lists fit): the epilogue then runs over them only
  {
    const int ipt = (nv + nt - 1) / nt;
    const int i0 = tid * ipt, i1 = min(nv, i0 + ipt);
    for (int v = i0; v < i1; v++) {
      float cx, cy;
      centroid(v, cx, cy);
      g_cen[v] = f32x2{cx, cy};  // computed once: the chunks and the epilogue read it back (a dense voxel has ~100 members and ~30 chunks)
      const Win w = window(cx, cy);
      int tot = 0;
      u32x4 rec = {0xFFFFFFFFu, 0u, 0u, 0u};  // the candidate ranges of a window of up to four voxel rows, for the sample's chunk lanes
      for (int gy = w.gy0; gy <= w.gy1; gy += 4) {
        int a[4], b[4];
        row_ranges4(w, gy, a, b);
        tot += (b[0] - a[0]) + (b[1] - a[1]) + (b[2] - a[2]) + (b[3] - a[3]);
        if (gy == w.gy0 && w.gy1 - w.gy0 < 4)
          rec = u32x4{(unsigned)a[0] | ((unsigned)b[0] << 16), (unsigned)a[1] | ((unsigned)b[1] << 16), (unsigned)a[2] | ((unsigned)b[2] << 16),
                      (unsigned)a[3] | ((unsigned)b[3] << 16)};
      }
      g_rng[v] = rec;
      ord[v] = (unsigned short)(tot >= 6 ? tot : 0);  // fewer than six candidates can never make a cell (pointnormal.cpp:291)
    }
    if (pt) pt->mark();
    CFEAR_STOP_AT(4, true);
    int o, oa;
    for (int it = 0;; it++) {  // block-uniform: the chunk list and the active-sample list side by side; if they do not fit, the chunk
                               // list alone (the epilogue then visits every sample), with the chunk size doubled until it fits
      int cnt = 0, act = 0;
      for (int i = i0; i < i1; i++) { const int t = (int)ord[i]; cnt += (t + C - 1) >> CS; act += t > 0 ? 1 : 0; }
      // one scan for the two counts: chunks in bits 0..18, active samples (<= nv <= 4864) in bits 19..31. A dense cloud with a
      // large downsample_factor has far more than 2^19 chunks of 16 (hundreds of voxels that each see thousands of candidates):
      // a thread's count is clamped to 1023 (512 x 1023 < 2^19, the fields cannot run into each other) and a clamped thread
      // raises a flag that travels with the partial sums - the chunk size is doubled then, as for any list that does not fit
      const bool big = cnt > 1023;
      cnt = big ? 1023 : cnt;
      unsigned tot2;
      bool any_big;
      const unsigned ex = block_exclusive_scan_1b_flag<CFEAR_FEAT_BLOCK>((unsigned)cnt | ((unsigned)act << 19), big, W.red_i, it & 1, &tot2, &any_big);
      o = (int)(ex & 0x7FFFFu); oa = (int)(ex >> 19); NC = (int)(tot2 & 0x7FFFFu); NA = (int)(tot2 >> 19);
      if (!any_big && NC + NA <= ccap) break;
      listed = false;
      if (!any_big && NC <= ccap) break;  // (an active sample has a chunk, so NC >= NA; NC -> NA <= nv <= ccap as the chunks grow)
      C <<= 1; CS++;
    }
    for (int i = i0; i < i1; i++) {  // chunk start per sample (over its candidate total), sample per chunk, active samples in order
      const int t = (int)ord[i];
      const int c = (t + C - 1) >> CS;
      ord[i] = (unsigned short)o;
      for (int j = 0; j < c; j++) chk[o + j] = (unsigned short)i;
      o += c;
      if (listed & (t > 0)) { chk[NC + oa] = (unsigned short)i; oa++; }
    }
    if (tid == 0) ord[nv] = (unsigned short)NC;
    __syncthreads();
  }
  if (pt) pt->mark();
  CFEAR_STOP_AT(5, true);
  const size_t cs = (size_t)W.cap;
  // One chunk per lane and trip; the lanes of a 16-lane row that hold chunks of the same sample (chunks are listed sample by
  // sample) then add their partial moments together with row shifts, and only the first lane of such a run stores: a dense
  // sample has ~30 chunks, and the epilogue's walk over them - a dependent memory round trip each - was its long pole.
  f32x2 cen_next = g_cen[tid < NC ? (int)chk[tid] : 0];
  u32x4 rec_next = g_rng[tid < NC ? (int)chk[tid] : 0];
  for (int wb = 0; wb < NC; wb += nt) {  // wave-uniform trip count: every lane takes part in the row shifts
    const int wq = wb + tid;
    const bool act = wq < NC;
    const int v = (int)chk[act ? wq : 0];
    const int j = wq - (int)ord[v];
    const float cx = cen_next.x, cy = cen_next.y;
    const u32x4 rec = rec_next;
    {  // the next chunk's centroid and candidate ranges are on their way while this one is summed
      const int vn = wq + nt < NC ? (int)chk[wq + nt] : 0;
      cen_next = g_cen[vn]; rec_next = g_rng[vn];
    }
    const bool stored = rec.x != 0xFFFFFFFFu;  // (a range never ends at 65535: at most CFEAR_CPT_CAP points)
    Win win = {0, 0, 0, 0};
    if (!stored) win = window(cx, cy);  // more than four voxel rows (leaf < radius): the ranges are looked up again
    int skip = j * C, left = act ? C : 0;
    int m = 0;
    double s0 = 0, s1x = 0, s1y = 0, sxx = 0, sxy = 0, syy = 0;
    const double cxd = (double)cx, cyd = (double)cy;
    // one candidate: branch-free - outside the radius (or past the chunk) it enters with weight 0: adding zeros leaves the sums
    // bit-identical, and the lanes of a wave hold different candidates anyway, so a branch would run its body for almost every trip
    auto add = [&](const f32x2 p, int iw, bool on) {
      const float dx = cx - p.x, dy = cy - p.y;
      float d2 = dx * dx; d2 += dy * dy;
      const bool in = on & (d2 < r2);  // pointnormal.cpp:291 radius test (float, strict)
      const double w = in ? (double)iw : 0.0;  // the weight byte staged with the point (:15)
      const double ex = (double)p.x - cxd, ey = (double)p.y - cyd;
      const double wex = w * ex, wey = w * ey;
      m += in ? 1 : 0; s0 += w; s1x += wex; s1y += wey;
      sxx = __builtin_fma(wex, ex, sxx); sxy = __builtin_fma(wex, ey, sxy); syy = __builtin_fma(wey, ey, syy);
    };
    if (stored) {
      // the sample's candidates numbered 0 .. tot - 1 through its (up to four) row ranges: candidate t of row u sits at t + off[u].
      // The chunk takes its 16 in four full trips, whatever the rows look like (row by row, ragged ends cost a trip each and
      // the lanes of a wave wait for the chunk with the most of them)
      const int a0 = (int)(rec.x & 0xFFFFu), a1 = (int)(rec.y & 0xFFFFu), a2 = (int)(rec.z & 0xFFFFu), a3 = (int)(rec.w & 0xFFFFu);
      const int p1 = (int)(rec.x >> 16) - a0, p2 = p1 + (int)(rec.y >> 16) - a1, p3 = p2 + (int)(rec.z >> 16) - a2, tot = p3 + (int)(rec.w >> 16) - a3;
      const int o1 = a1 - p1, o2 = a2 - p2, o3 = a3 - p3;
      const int t0 = act ? skip : 0, t1 = act ? min(skip + C, tot) : 0;
      for (int t = t0; t < t1; t += 4) {
        f32x2 p[4]; int iw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {  // independent LDS loads first
          const int tt = min(t + u, t1 - 1);
          const int qq = tt + (tt < p1 ? a0 : (tt < p2 ? o1 : (tt < p3 ? o2 : o3)));
          p[u] = pxy[qq]; iw[u] = (int)pw[qq];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) add(p[u], iw[u], t + u < t1);
      }
    } else
    for (int gy4 = win.gy0; gy4 <= win.gy1 && left > 0; gy4 += 4) {  // more than four voxel rows: row by row
     int ra[4], rb[4];
     row_ranges4(win, gy4, ra, rb);
#pragma unroll
     for (int u = 0; u < 4; u++) {
      const int a = ra[u], b = rb[u];
      const int len = b - a;
      if (left <= 0 || len <= 0) continue;
      if (skip >= len) { skip -= len; continue; }
      const int s = a + skip, e = min(b, s + left);
      for (int q = s; q < e; q += 4) {
        f32x2 p[4]; int iw[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const int qq = min(q + u, e - 1); p[u] = pxy[qq]; iw[u] = (int)pw[qq]; }  // independent LDS loads first
#pragma unroll
        for (int u = 0; u < 4; u++) add(p[u], iw[u], q + u < e);
      }
      left -= e - s; skip = 0;
     }
    }
    double acc[7] = {(double)m, s0, s1x, s1y, sxx, sxy, syy};
    const int key = act ? v + 1 : 0;
#define CFEAR_ROW_STEP(D)                                                                                   \
    {                                                                                                       \
      const bool same = act & (__builtin_amdgcn_update_dpp(0, key, 0x100 + (D), 0xF, 0xF, true) == key);    \
      _Pragma("unroll") for (int q = 0; q < 7; q++) {                                                       \
        const double t = dpp_get<0x100 + (D), 0xF>(acc[q]); /* row_shl: the value of lane + D of this row, 0 past its end */ \
        acc[q] += same ? t : 0.0;                                                                           \
      }                                                                                                     \
    }
    CFEAR_ROW_STEP(1) CFEAR_ROW_STEP(2) CFEAR_ROW_STEP(4) CFEAR_ROW_STEP(8)
#undef CFEAR_ROW_STEP
    const bool head = ((tid & 15) == 0) | (__builtin_amdgcn_update_dpp(0, key, 0x111, 0xF, 0xF, true) != key);  // row_shr:1 = lane - 1
    if (act & head) {
#pragma unroll
      for (int q = 0; q < 7; q++) g_part[q * cs + wq] = acc[q];
    }
  }
  __syncthreads();
  if (pt) pt->mark();
  CFEAR_STOP_AT(6, true);
  // ---- cell epilogue + compaction: a cell is built in registers and, if it is valid, written straight to its final
  // slot; one block scan per round of blockDim samples keeps the sample order (pointnormal.cpp:292-294)
  int n_cells_out;
  // float cell means for the grid build go over the bitmap and its prefix (the rank queries are done): no read-back from memory
  constexpr int LM_CAP = (int)((FeatLdsC::vst - FeatLdsC::bm) / 8);
  typedef __attribute__((address_space(3))) float l_f32;
  l_f32* const lmw = (l_f32*)(lds + FeatLdsC::bm);
  {
    int base = 0;
    const int cap_cells = S->cap_cells;
    const bool keep_cells = S->cells != nullptr;
    const int NE = listed ? NA : nv;
    for (int a0 = 0, round = 0; a0 < NE; a0 += nt, round++) {  // rounds over the active (or all) samples, in sample order
      const int ai = a0 + tid;
      cfear_cell c;
      int valid = 0;
      if (ai < NE) {
        const int v = listed ? (int)chk[NC + ai] : ai;
        double md = 0, s0 = 0, s1x = 0, s1y = 0, sxx = 0, sxy = 0, syy = 0;
        const int w0 = (int)ord[v], w1 = (int)ord[v + 1];
        // the sample's partial sums sit at its first chunk and at every chunk that starts a 16-lane row (see above); two per
        // trip: fourteen loads in flight together, added in chunk order
        for (int w = w0; w < w1;) {
          const int wn = (w | 15) + 1, wb = min(wn, w1 - 1);
          double pa[7], pb[7];
#pragma unroll
          for (int q = 0; q < 7; q++) { pa[q] = g_part[q * cs + w]; pb[q] = g_part[q * cs + wb]; }
          md += pa[0]; s0 += pa[1]; s1x += pa[2]; s1y += pa[3]; sxx += pa[4]; sxy += pa[5]; syy += pa[6];
          if (wn < w1) { md += pb[0]; s0 += pb[1]; s1x += pb[2]; s1y += pb[3]; sxx += pb[4]; sxy += pb[5]; syy += pb[6]; }
          w = (wn | 15) + 1;
        }
        const int m = (int)md;
        if (m >= 6) {  // :291
          const f32x2 cen = g_cen[v];
          const float cx = cen.x, cy = cen.y;
          const double is0 = 1.0 / s0;  // one division for the five quotients by the weight sum (an ulp or two away from five divisions)
          const double m1x = s1x * is0, m1y = s1y * is0;
          const double ux = (double)cx + m1x, uy = (double)cy + m1y;
          const double cxx = sxx * is0 - m1x * m1x, cyx = sxy * is0 - m1x * m1y, cyy = syy * is0 - m1y * m1y;
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
      const int o = base + block_exclusive_scan_1b<CFEAR_FEAT_BLOCK>(valid, W.red_i, round & 1, &round_total);
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
        if (o < LM_CAP) { lmw[2 * o] = (float)c.mean[0]; lmw[2 * o + 1] = (float)c.mean[1]; }
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
  if (pt) { pt->mark(); pt->mark(); }
  CFEAR_STOP_AT(7, true);
  // ---- uniform grid over the float cell means (replaces KdTreeFLANN<PointXY>, :151-162): counters and offsets go over
  // the staged points (consumed)
  FeatureScratch Wg = W;
  Wg.keys = reinterpret_cast<uint64_t*>(lds + FeatLdsC::pxy);
  Wg.tab_voxels = (int)(CFEAR_CPT_CAP * 8 / 2);  // ints that fit the region x 2 (cell_grid_block's unit: 16-bit counters)
  Wg.vlist = reinterpret_cast<int*>(lds + FeatLdsC::bm);
  Wg.lds = true;
  cell_grid_block(S, n_cells_out, P, Wg, n_cells_out <= LM_CAP, pt);
  return true;
}

}  // namespace cfear_dev
