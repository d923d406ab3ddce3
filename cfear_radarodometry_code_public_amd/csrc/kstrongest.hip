// kstrongest.hip -- K1/K2: batched k-strongest filter + axial non-max suppression for gfx950.
//
// Replaces StructuredKStrongest::FilterKstrongest (radar_filters.cpp:209-237) and
// AxialNonMaxSupress (radar_filters.cpp:238-298) for n_scans x A azimuth rows of R uint8 bins.
//
// Mapping: one 64-lane wavefront owns one azimuth row (persistent wave loop over rows). The row is
// streamed from HBM as aligned 16-byte chunks, chunk c = j*64 + lane, i.e. every load instruction of
// the wave covers 1 KiB contiguous. The chunks stay in VGPRs for the selection and are also staged
// in a wave-private LDS window (row + 6-byte halo either side) for the byte lookups of the
// compaction and of the 13-tap non-max-suppression window.
//
// Selection (bit-exact with the reference's bounded sorted insert): the result is the k largest
// keys (intensity, range) among bytes >= z_min. Instead of sorting, the wave
//   1. probes the previous azimuth's threshold T (a wave walks consecutive rows); if that yields fewer than k
//      candidates it bounds the k-th largest intensity from below by the k-th largest per-lane maximum,
//   2. counts bytes >= T with SWAR compares (4 bytes / 5 VALU ops) and narrows T by bisection
//      until k <= count <= 64 (usually the first probe),
//   3. gives every candidate a lane without a per-candidate loop (prefix sums, owner scatter + DPP max-scan,
//      popcount split + byte table), ranks the keys with LDS broadcast reads (the key includes the range, so
//      ties go to the larger range bin exactly like std::pair<uchar,int> ordering) and
//   4. handles rows with > 64 ties at the threshold intensity by a backwards positional scan.
// No block-level barrier is used: the four waves of a block are independent.
#include <stdlib.h>

#include "common.h"

// tools/build_k1_stop_variants.sh: profile builds whose row loop ends after phase n (1 load + LDS staging, 2 threshold search and
// candidate masks, 3 candidates to lanes, 4 ranking; 0 = the product kernel); PMC counters of successive variants difference into
// per-phase instruction counts (tools/pmc_k1_phases.sh). What a stopped row stores depends on the phase's results, so nothing of
// the phase is optimised away.
#ifndef CFEAR_K1_STOP
#define CFEAR_K1_STOP 0
#endif
#define K1_STOP_AT(n, val)                                                           \
  if (CFEAR_K1_STOP == (n)) {                                                        \
    if (lane < k) slots[g * (long long)k + lane] = (uint32_t)(val);                  \
    wave_lds_fence();                                                                \
    continue;                                                                        \
  }

namespace {

constexpr uint32_t HI = 0x80808080u;

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave64 inclusive prefix sum with DPP adds (Hillis-Steele inside the 16-lane rows, then two row broadcasts)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_add_i(int v) {
  return v + __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);  // lanes without a source add 0
}
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v = dpp_add_i<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add_i<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add_i<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add_i<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add_i<0x142, 0xA>(v);  // row_bcast15 -> rows 1, 3
  v = dpp_add_i<0x143, 0xC>(v);  // row_bcast31 -> rows 2, 3
  return v;
}

// same shape with max (values >= 0): lane i gets the maximum over lanes 0..i
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_max_i(int v) {
  const int o = __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
  return o > v ? o : v;
}
__device__ __forceinline__ int wave_inclusive_max(int v) {
  v = dpp_max_i<0x111, 0xF>(v);
  v = dpp_max_i<0x112, 0xF>(v);
  v = dpp_max_i<0x114, 0xF>(v);
  v = dpp_max_i<0x118, 0xF>(v);
  v = dpp_max_i<0x142, 0xA>(v);
  v = dpp_max_i<0x143, 0xC>(v);
  return v;
}

// ... with OR: lane 63 ends up with the OR over the wave
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_or_i(int v) {
  return v | __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int wave_inclusive_or(int v) {
  v = dpp_or_i<0x111, 0xF>(v);
  v = dpp_or_i<0x112, 0xF>(v);
  v = dpp_or_i<0x114, 0xF>(v);
  v = dpp_or_i<0x118, 0xF>(v);
  v = dpp_or_i<0x142, 0xA>(v);
  v = dpp_or_i<0x143, 0xC>(v);
  return v;
}

// wave-wide sum of a small non-negative per-lane integer: six DPP adds and a readlane (the ballot bit-slicing
// it replaces cost two VALU instructions per bit)
template <int BITS>
__device__ __forceinline__ int wave_sum_small(int v) {
  return __builtin_amdgcn_readlane(wave_inclusive_scan(v), 63);
}

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// bytes [lo, hi) of a 16-byte chunk kept, the rest zeroed
__device__ __forceinline__ uint32_t dword_keep(int lo, int hi) {
  lo = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
  hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
  if (hi <= lo) return 0u;
  const uint32_t mh = hi == 4 ? 0xFFFFFFFFu : ((1u << (8 * hi)) - 1u);
  const uint32_t ml = lo == 0 ? 0u : ((1u << (8 * lo)) - 1u);
  return mh & ~ml;
}
__device__ __forceinline__ uint4 chunk_keep(uint4 v, int lo, int hi) {
  if (lo <= 0 && hi >= 16) return v;
  v.x &= dword_keep(lo, hi);
  v.y &= dword_keep(lo - 4, hi - 4);
  v.z &= dword_keep(lo - 8, hi - 8);
  v.w &= dword_keep(lo - 12, hi - 12);
  return v;
}

// per-byte (x >= T) in bit 7 of every byte (the other bits are not cleared); brep = (T & 0x7f) replicated, THIGH = (T >= 128)
template <bool THIGH>
__device__ __forceinline__ uint32_t ge_bit7(uint32_t x, uint32_t brep) {
  const uint32_t d = (x | HI) - brep;
  return THIGH ? (x & d) : (x | d);
}
__device__ __forceinline__ uint32_t bfi(uint32_t mask, uint32_t a, uint32_t b) { return (a & mask) | (b & ~mask); }  // v_bfi_b32

// Chunk candidate mask layout: bit (8*b + d) <-> byte b of dword d (byte index 4*d + b in the chunk). The four flag bits of a
// byte position are gathered with bit-field inserts (bit 7 from dword 3, bit 6 from dword 2 >> 1, ...: one v_bfi_b32 per dword
// instead of an AND with 0x80808080 and an OR) and moved down once.
template <int NCH, bool THIGH>
__device__ __forceinline__ void ge_masks_t(const uint4 (&v)[NCH], uint32_t brep, uint32_t (&m)[NCH]) {
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    const uint32_t t0 = ge_bit7<THIGH>(v[j].x, brep), t1 = ge_bit7<THIGH>(v[j].y, brep);
    const uint32_t t2 = ge_bit7<THIGH>(v[j].z, brep), t3 = ge_bit7<THIGH>(v[j].w, brep);
    uint32_t u = bfi(0x80808080u, t3, t2 >> 1);
    u = bfi(0xC0C0C0C0u, u, t1 >> 2);
    u = bfi(0xE0E0E0E0u, u, t0 >> 3);
    m[j] = (u >> 4) & 0x0F0F0F0Fu;
  }
}
// masks of the bytes >= T (1 <= T <= 255; T == 256 -> none), restricted to the row by the validity
// masks of the first chunk group (vhead, j == 0) and of the chunk groups >= jt (vtail); returns the
// per-lane candidate count
template <int NCH>
__device__ __forceinline__ int count_mask(const uint4 (&v)[NCH], int T, uint32_t (&m)[NCH], uint32_t vhead, int jt,
                                          uint32_t vtail, uint32_t vtail2) {
  if (T >= 256) {
#pragma unroll
    for (int j = 0; j < NCH; j++) m[j] = 0;
    return 0;
  }
  const uint32_t brep = (uint32_t)(T & 0x7f) * 0x01010101u;
  if (T >= 128) ge_masks_t<NCH, true>(v, brep, m); else ge_masks_t<NCH, false>(v, brep, m);
  m[0] &= vhead;
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    if (j == jt) m[j] &= vtail;
    if (j == jt + 1) m[j] &= vtail2;
    if (j > jt + 1) m[j] = 0;
    cnt += __popc(m[j]);
  }
  return cnt;
}

// per-lane maximum row byte, from the registers the row is held in (the cold / cluttered-row paths). Only the first chunk group and
// the group holding the row end can have bytes outside the row; their validity flags (candidate-mask layout: bit 8b + d <-> byte b
// of dword d) are widened to byte masks - t = the flags of dword d in bit 0 of every byte, (t << 8) - t = 0xFF in every flagged
// byte - and the rest is v_pk_max_u16 on the odd bytes and on the even bytes << 8. (Until round 4 the row was read back from LDS
// and every chunk masked by byte ranges: twice the instructions.)
template <int NCH>
__device__ __forceinline__ int lane_max_byte_regs(const uint4 (&v)[NCH], uint32_t vhead, int jt, uint32_t vtail) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  us2 ao = {0, 0}, ae = {0, 0};
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
    if (j == 0 || j >= jt) {  // (jt is wave-uniform)
      const uint32_t f = j > jt ? 0u : (j == jt ? vtail : vhead);  // (vtail includes vhead when the row ends in the first group)
#pragma unroll
      for (int d = 0; d < 4; d++) {
        const uint32_t t = (f >> d) & 0x01010101u;
        w[d] &= (t << 8) - t;
      }
    }
#pragma unroll
    for (int d = 0; d < 4; d++) {
      ao = __builtin_elementwise_max(ao, __builtin_bit_cast(us2, w[d]));
      ae = __builtin_elementwise_max(ae, __builtin_bit_cast(us2, w[d] << 8));
    }
  }
  const int m0 = ao.x >> 8, m1 = ao.y >> 8, m2 = ae.x >> 8, m3 = ae.y >> 8;
  const int a = m0 > m1 ? m0 : m1, b = m2 > m3 ? m2 : m3;
  return a > b ? a : b;
}

// per-lane candidate mask of all chunks: chunk j occupies the bit set {8b + d + 4*(j&1)} of word j>>1
// (b = byte in dword, d = dword in chunk), i.e. two chunks interleave into one 32-bit word.
template <int NCH>
__device__ __forceinline__ void pack_masks(const uint32_t (&m)[NCH], uint32_t (&w)[NCH / 2]) {
#pragma unroll
  for (int i = 0; i < NCH / 2; i++) w[i] = m[2 * i] | (m[2 * i + 1] << 4);
}

// candidate-mask bits (layout 8*b + d) of the chunk bytes with index in [lo_b, hi_b)
__device__ __forceinline__ uint32_t chunk_range_mask(int lo_b, int hi_b) {
  uint32_t keep = 0;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    int l = lo_b - 4 * d, h = hi_b - 4 * d;
    l = l < 0 ? 0 : (l > 4 ? 4 : l);
    h = h < 0 ? 0 : (h > 4 ? 4 : h);
    const uint32_t nib = h > l ? (((1u << h) - 1u) & ~((1u << l) - 1u)) : 0u;  // bits b in [l, h)
    const uint32_t spread = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
    keep |= spread << d;
  }
  return keep;
}

// backwards positional scan over the LDS row: position of the need-th largest range bin whose byte == val
__device__ __forceinline__ int tie_position(const uint8_t* win, int head, int R, int val, int need, int lane) {
  int acc = 0;
  for (int tb = R - 64; tb > -64; tb -= 64) {
    const int pos = tb + lane;
    const bool hit = pos >= 0 && win[head + pos] == (uint8_t)val;
    unsigned long long bb = __ballot(hit);
    const int c = __popcll(bb);
    if (acc + c >= need) {
      int want = need - acc, bit = 63;
      while (true) {
        bit = 63 - __clzll(bb);
        if (--want == 0) break;
        bb &= ~(1ull << bit);
      }
      return tb + bit;
    }
    acc += c;
  }
  return 0;
}

template <int NCH, int OCC>
__global__ __launch_bounds__(256, OCC) void kstrongest_kernel(const uint8_t* __restrict__ polar,
                                                            uint32_t* __restrict__ slots, int A, int R,
                                                            long long n_rows, int u_zmin, int k,
                                                            long long alloc_bytes, int rows_per_wave) {
  constexpr int WIN = NCH * 1024;  // LDS window bytes per wave
  constexpr int SUMBITS = NCH == 4 ? 7 : (NCH == 8 ? 8 : 9);
  __shared__ uint4 lds_win[4][NCH * 64];
  __shared__ __attribute__((aligned(16))) uint32_t lds_keys[4][64];
  __shared__ uint32_t lds_ge[20];  // lds_ge[n] = candidate-mask bits of the chunk bytes with index >= n (n = 0..16)
  __shared__ uint32_t lds_cmp[4][(1 + NCH / 2) * 64];  // per wave: slot offsets and candidate-mask words of every lane
  __shared__ uint8_t lds_nth[256 * 8];                 // lds_nth[b * 8 + r] = position of the r-th set bit of byte b
  if (threadIdx.x <= 16) lds_ge[threadIdx.x] = chunk_range_mask((int)threadIdx.x, 16);
  {
    const uint32_t b = threadIdx.x;  // 256 threads, one byte value each
    int r = 0;
    for (int p = 0; p < 8; p++)
      if ((b >> p) & 1u) lds_nth[b * 8 + r++] = (uint8_t)p;
    for (; r < 8; r++) lds_nth[b * 8 + r] = 0;
  }
  __syncthreads();  // the only block-level barrier: before the waves go their own way
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // scalar: keeps all per-row bookkeeping on the SALU
  uint8_t* const win = reinterpret_cast<uint8_t*>(&lds_win[wave][0]);
  uint32_t* const keys = &lds_keys[wave][0];
  uint32_t* const cmp_ex = &lds_cmp[wave][0];   // [64] first slot of every lane
  uint32_t* const cmp_w = &lds_cmp[wave][64];   // [NCH/2][64] mask words of every lane
  const long long scan_bytes = (long long)A * (long long)R;
  const int Tfloor = u_zmin > 1 ? u_zmin : 1;

  // each wave owns rows_per_wave consecutive rows: the selection threshold of one azimuth is the
  // first guess for the next one
  const long long g0 = ((long long)blockIdx.x * 4 + wave) * rows_per_wave;
  long long g1 = g0 + rows_per_wave;
  if (g1 > n_rows) g1 = n_rows;
  int Tprev = Tfloor;

  long long scan = g0 / A;
  int bearing = (int)(g0 - scan * A);
  for (long long g = g0; g < g1; g++, bearing++) {
    if (bearing == A) { bearing = 0; scan++; }
    // all addressing relative to the (16-byte aligned) kernel argument so that the loads are
    // global_load_dwordx4 with a scalar base and a 32-bit lane offset
    const long long row_off = g * (long long)R;
    const long long wstart_off = (row_off - 6) & ~15LL;     // may be -16 for the very first row
    const int head = (int)(row_off - wstart_off);           // 6..21: window offset of range bin 0
    // chunks [c_lo, c_hi) are inside the allocation and needed (row + 6-byte halo either side)
    const int c_lo = wstart_off >= 0 ? 0 : (int)((-wstart_off + 15) >> 4);
    int c_hi = (head + R + 6 + 15) >> 4;
    {
      const long long lim = (alloc_bytes - wstart_off + 15) >> 4;
      if (lim < c_hi) c_hi = (int)lim;
    }
    const bool edge_row = bearing == 0 || bearing == A - 1;
    const uint8_t* const wp = polar + wstart_off;  // wave-uniform

    // ---- load: HBM -> VGPR and LDS (scan-masked on the first/last row, keeps the cross-row halo).
    // Branch-free: out-of-range chunks read a clamped (valid) address and are zeroed afterwards, so
    // the NCH loads of 1 KiB each are all in flight together.
    uint4 v[NCH];
    if (!edge_row && c_lo == 0) {
      // common case (398 of 400 azimuths): everything the window needs lies inside this scan's image. Chunks past
      // the halo re-read the last needed chunk (no extra HBM traffic); what they hold is never looked at: the
      // selection masks them (vtail) and the suppression reads at most 6 bytes past the row.
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int cc = min(j * 64 + lane, c_hi - 1);
        {
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wp + (uint32_t)(16 * cc)));
          v[j] = make_uint4(t.x, t.y, t.z, t.w);
        }
      }
#pragma unroll
      for (int j = 0; j < NCH; j++) reinterpret_cast<uint4*>(win)[j * 64 + lane] = v[j];
    } else {
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      const int c = j * 64 + lane;
      const int cc = min(max(c, c_lo), c_hi - 1);
      v[j] = *reinterpret_cast<const uint4*>(wp + (uint32_t)(16 * cc));
    }
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      const int c = j * 64 + lane;
      if (c < c_lo || c >= c_hi) v[j] = make_uint4(0, 0, 0, 0);
      uint4 staged = v[j];
      if (edge_row) {
        // bytes outside this scan's image read as 0 (the reference's unchecked cv::Mat::at would run
        // off the buffer there, radar_filters.cpp:260)
        const long long ca = wstart_off + 16LL * c;
        const long long slo = scan * scan_bytes - ca, shi = (scan + 1) * scan_bytes - ca;
        if (slo > 0 || shi < 16) staged = chunk_keep(staged, (int)(slo > 16 ? 16 : slo), (int)(shi < 0 ? 0 : (shi > 16 ? 16 : shi)));
      }
      reinterpret_cast<uint4*>(win)[c] = staged;
    }
    }
    // validity of the chunk bytes w.r.t. the row [0, R): only the first chunk group and the group(s)
    // holding the row end can be partial
    const int jt = (head + R - 1) >> 10;  // chunk group of the last range bin
    int hb = head - 16 * lane, tb = head + R - 16 * (jt * 64 + lane);
    hb = hb < 0 ? 0 : (hb > 16 ? 16 : hb);
    tb = tb < 0 ? 0 : (tb > 16 ? 16 : tb);
    const uint32_t vhead = lds_ge[hb];
    const uint32_t vtail = (~lds_ge[tb] & 0x0F0F0F0Fu) & (jt == 0 ? vhead : 0x0F0F0F0Fu);
    const uint32_t vtail2 = 0u;  // groups after jt hold no row bytes
    wave_lds_fence();
    K1_STOP_AT(1, v[0].x ^ v[1].y ^ v[2].z ^ v[3].w ^ vhead ^ vtail);

    // ---- threshold search: first probe = previous row's threshold ----
    int lo = Tprev;
    uint32_t m[NCH];
    int cnt = wave_sum_small<SUMBITS>(count_mask<NCH>(v, lo, m, vhead, jt, vtail, vtail2));
    if (cnt < k && lo > Tfloor) {
      // too few: restart from the floor, bounded below by the k-th largest per-lane maximum
      lo = Tfloor;
      int lm = lane_max_byte_regs<NCH>(v, vhead, jt, vtail);
      if (lm < Tfloor) lm = 0;
      int tl = 0, th = 256;
      while (th - tl > 1) {
        const int mid = (tl + th) >> 1;
        if (__popcll(__ballot(lm >= mid)) >= k) tl = mid; else th = mid;
      }
      if (tl > lo) lo = tl;  // count(bytes >= tl) >= k is guaranteed
      cnt = wave_sum_small<SUMBITS>(count_mask<NCH>(v, lo, m, vhead, jt, vtail, vtail2));
    }
    if (cnt > 64 && lo < 255) {
      // far too many (a wave's first row starts from z_min; a cluttered row after a quiet one): jump to the k-th largest per-lane
      // maximum - at least k lanes hold a byte that large, so count(>= it) >= k - instead of bisecting with a full recount per
      // step (eight recounts on uniformly distributed bytes, where ~76 % of the bins pass z_min)
      int lm = lane_max_byte_regs<NCH>(v, vhead, jt, vtail);
      int tl = lo, th = 256;
      while (th - tl > 1) {
        const int mid = (tl + th) >> 1;
        if (__popcll(__ballot(lm >= mid)) >= k) tl = mid; else th = mid;
      }
      if (tl > lo) {
        lo = tl;
        cnt = wave_sum_small<SUMBITS>(count_mask<NCH>(v, lo, m, vhead, jt, vtail, vtail2));
      }
    }
    if (cnt > 64) {
      int hi = 256;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        uint32_t mm[NCH];
        const int c2 = wave_sum_small<SUMBITS>(count_mask<NCH>(v, mid, mm, vhead, jt, vtail, vtail2));
        if (c2 >= k) {
          lo = mid; cnt = c2;
#pragma unroll
          for (int j = 0; j < NCH; j++) m[j] = mm[j];
          if (c2 <= 64) break;
        } else {
          hi = mid;
        }
      }
    }
    // next row starts from this threshold (one higher when the candidate set is getting large)
    Tprev = (cnt > 40 && lo < 255) ? lo + 1 : lo;

    // ---- ties: > 64 bytes equal to the threshold intensity, or z_min == 0 and zeros are needed ----
    if (cnt > 64) {
      uint32_t mg[NCH];
      const int c_gt = wave_sum_small<SUMBITS>(count_mask<NCH>(v, lo + 1, mg, vhead, jt, vtail, vtail2));
      const int pstar = tie_position(win, head, R, lo, k - c_gt, lane);
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int bp = 16 * (j * 64 + lane) - head;  // range bin of byte 0 of this chunk
        const int nb = pstar - bp;                   // chunk bytes with index >= nb have range >= pstar (m is inside the row already)
        const uint32_t keep = lds_ge[nb < 0 ? 0 : (nb > 16 ? 16 : nb)];
        m[j] = mg[j] | (m[j] & ~mg[j] & keep);  // (> lo) | (== lo & range >= pstar)
      }
      cnt = k;
    } else if (u_zmin == 0 && cnt < k && R > cnt && lo == 1) {
      // z_min == 0: zero-valued bins are candidates too; take the largest ranges among them
      int need = k - cnt;
      if (need > R - cnt) need = R - cnt;
      const int pstar = tie_position(win, head, R, 0, need, lane);
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int bp = 16 * (j * 64 + lane) - head;
        const uint32_t keep = chunk_range_mask(pstar - bp, R - bp);  // (zeros are not in m: the row end has to be cut here; rare path)
        m[j] |= (~m[j]) & 0x0F0F0F0Fu & keep;
      }
      cnt += need;
    }

    K1_STOP_AT(2, (uint32_t)cnt + (m[0] ^ m[1] ^ m[NCH - 2] ^ m[NCH - 1]));
    // ---- compaction without a per-candidate loop (strong returns cluster: a lane often holds five or more
    // candidates). Lane l's candidates take the slots ex[l] .. ex[l] + count - 1 (DPP prefix sum). The lane that
    // owns slot s finds its producer l (producers scatter their id to their first slot, a DPP max-scan spreads it),
    // fetches l's mask words from LDS and picks set bit number s - ex[l] (popcount split + byte table) ----
    const int C = cnt < 64 ? cnt : 64;
    const int kk = k < C ? k : C;  // number of emitted points
    uint32_t key = 0u;
    {
      uint32_t w[NCH / 2];
      pack_masks<NCH>(m, w);
      int c = 0;
#pragma unroll
      for (int i = 0; i < NCH / 2; i++) c += __popc(w[i]);
      const int ex = wave_inclusive_scan(c) - c;
      keys[lane] = 0u;
      cmp_ex[lane] = (uint32_t)ex;
#pragma unroll
      for (int i = 0; i < NCH / 2; i++) cmp_w[i * 64 + lane] = w[i];
      wave_lds_fence();
      if (c > 0 && ex < 64) keys[ex] = (uint32_t)(lane + 1);
      wave_lds_fence();
      const int owner = wave_inclusive_max((int)keys[lane]) - 1;
      if (lane < C) {
        const int l = owner;
        int r = lane - (int)cmp_ex[l];
        uint32_t sel = 0u;
        int wi = 0;
        bool found = false;
#pragma unroll
        for (int i = 0; i < NCH / 2; i++) {
          const uint32_t wv = cmp_w[i * 64 + l];
          const int ci = __popc(wv);
          if (!found) {
            if (r < ci) { sel = wv; wi = i; found = true; } else r -= ci;
          }
        }
        int t = 0;  // position of set bit number r of sel
        {
          const uint32_t lo16 = sel & 0xFFFFu;
          const int c16 = __popc(lo16);
          if (r >= c16) { r -= c16; t = 16; sel >>= 16; } else sel = lo16;
          const uint32_t lo8 = sel & 0xFFu;
          const int c8 = __popc(lo8);
          if (r >= c8) { r -= c8; t += 8; sel >>= 8; } else sel = lo8;
          t += lds_nth[sel * 8 + r];
        }
        const int j = 2 * wi + ((t >> 2) & 1);
        const int bi = 4 * (t & 3) + (t >> 3);  // byte index inside the chunk
        const uint32_t woff = (uint32_t)(16 * (j * 64 + l) + bi);  // window offset
        key = (woff - (uint32_t)head) | ((uint32_t)win[woff] << 16) | (1u << 24);
      }
    }
    K1_STOP_AT(3, key);
    // ---- ranks. Few candidates (C <= 2k, the usual case behind a threshold that tracks the previous azimuth): every lane counts the
    // larger keys among all C - four keys per LDS instruction (broadcast reads), a compare and an add per key; lanes >= C hold key 0,
    // so reading past C is harmless. Many candidates (a cold threshold, uniformly distributed bytes: C up to 64): two vector
    // instructions per key are the largest item of the row, so first K* = the k-th largest key by bisection on count(key >= mid) - one
    // vector compare per step, the count and the interval on the scalar unit, ending as soon as a step counts exactly k (keys are
    // distinct: the range bin is part of the key, so equal intensities are split by range exactly like std::pair<uchar, int>) -
    // then the k kept keys move to the front of the LDS array (a prefix count of the ballot) and are ranked among themselves.
    bool kept;
    int rank = 0;
    wave_lds_fence();  // every lane has fetched its slot code
    if (C > 2 * k) {  // wave-uniform
      const uint32_t k24 = key & 0xFFFFFFu;  // intensity << 16 | range (lanes >= C hold 0)
      uint32_t klo = (uint32_t)lo << 16, khi = 256u << 16;  // count(>= klo) = C > k, count(>= khi) = 0
      while (khi - klo > 1u) {
        const uint32_t mid = (klo + khi) >> 1;
        const int c = __popcll(__ballot(k24 >= mid));
        if (c >= k) { klo = mid; if (c == k) break; } else khi = mid;
      }
      kept = k24 >= klo;  // exactly k lanes (kk = k)
      const unsigned long long keptb = __ballot(kept);
      const int kpos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(keptb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)keptb, 0u));
      if (kept) keys[kpos] = key;
      if (lane >= kk && lane < kk + 4) keys[lane] = 0u;  // the rest of the last group of four (key 0 is smaller than any kept key)
      wave_lds_fence();
      for (int j = 0; j < kk; j += 4) {
        const uint4 ka = reinterpret_cast<const uint4*>(keys)[j >> 2];
        rank += (ka.x > key) ? 1 : 0; rank += (ka.y > key) ? 1 : 0; rank += (ka.z > key) ? 1 : 0; rank += (ka.w > key) ? 1 : 0;
      }
    } else {
      keys[lane] = key;
      wave_lds_fence();
      for (int j = 0; j < C; j += 4) {
        const uint4 ka = reinterpret_cast<const uint4*>(keys)[j >> 2];
        rank += (ka.x > key) ? 1 : 0; rank += (ka.y > key) ? 1 : 0; rank += (ka.z > key) ? 1 : 0; rank += (ka.w > key) ? 1 : 0;
      }
      kept = lane < C && rank < kk;
    }
    const int mpos = (int)(key & 0xFFFFu);
    K1_STOP_AT(4, key + (uint32_t)rank);

    // ---- axial non-max suppression (radar_filters.cpp:238-298) on the kept points ----
    uint32_t peak = 0;
    const bool interior = mpos >= 3 && mpos < R - 3;  // :251
    uint32_t covered = 0x7Fu;
    const bool edge_points = __ballot(kept && !interior) != 0;  // wave-uniform: a kept point within 3 bins of a row end (rare)
    if (edge_points) {
      // scores exist only where a valid kept point's +-3 window put them in the map (:253-263); everything else reads as the
      // unordered_map default 0 (:271-276). An interior point's own window covers its seven scores; a point within three bins of a
      // row end only sees what INTERIOR kept points within six bins of that end wrote: positions 3 .. 8 (and R - 9 .. R - 4). Those are
      // marked with one bit each, the marks of the wave OR-ed by six DPP steps, spread over their +-3 windows by three shift-ors
      // (E bit j <-> a score exists at bin j - 3, resp. R - 12 + j), and a lane shifts its seven bits out. (It was a loop over every kept
      // lane with seven compare-ors each, in every wave that holds an edge point: with tied intensities the kept points crowd at the
      // far end of the row - ties go to the larger range - and every row took it.)
      if (R >= 24) {
        const bool ik = kept && interior;
        int marks = (ik && mpos <= 8) ? (1 << (mpos - 3)) : 0;
        marks |= (ik && mpos >= R - 9) ? (1 << (6 + mpos - (R - 9))) : 0;
        marks = __builtin_amdgcn_readlane(wave_inclusive_or(marks), 63);
        uint32_t es = (uint32_t)marks & 0x3Fu, ee = ((uint32_t)marks >> 6) & 0x3Fu;
        es |= es << 1; es |= es << 2; es |= es << 3;  // bit b set -> bits b .. b + 6
        ee |= ee << 1; ee |= ee << 2; ee |= ee << 3;
        const uint32_t c_lo = ((es << 3) >> (mpos < 3 ? mpos : 0)) & 0x7Fu;          // bin r = mpos - 3 + t <-> bit r + 3 of es << 3
        const uint32_t c_hi = (ee >> (mpos >= R - 3 ? mpos - R + 9 : 0)) & 0x7Fu;    // <-> bit r - (R - 12) of ee
        covered = interior ? 0x7Fu : (mpos < 3 ? c_lo : c_hi);
      } else {  // a row too short for the two ends to be apart: the plain loop
        covered = 0;
        unsigned long long kb = __ballot(kept);
        while (kb) {
          const int i = __ffsll((long long)kb) - 1;
          kb &= kb - 1;
          const int mi = (int)(__builtin_amdgcn_readlane((int)key, i) & 0xFFFF);
          if (mi >= 3 && mi < R - 3) {
#pragma unroll
            for (int t = 0; t < 7; t++) {
              const int r = mpos - 3 + t;
              if (r >= mi - 3 && r <= mi + 3) covered |= 1u << t;
            }
          }
        }
      }
    }
    if (kept) {
      const int off0 = head + mpos - 6;
      int bv[13];
#pragma unroll
      for (int t = 0; t < 13; t++) {
        bv[t] = win[off0 + t];  // 0 <= off0 + t < WIN: head >= 6 and R + 27 <= WIN
      }
      int sm[7];  // sm[u] = sum bv[u..u+6] (7-tap box, radar_filters.cpp:258-261)
      sm[0] = ((bv[0] + bv[1]) + (bv[2] + bv[3])) + ((bv[4] + bv[5]) + bv[6]);
#pragma unroll
      for (int u = 1; u < 7; u++) sm[u] = sm[u - 1] - bv[u - 1] + bv[u + 6];
      bool largest = true;
      if (edge_points) {  // (all seven scores exist otherwise: the masking is skipped for the whole wave)
#pragma unroll
        for (int u = 0; u < 7; u++)
          if (!((covered >> u) & 1u)) sm[u] = 0;
      }
#pragma unroll
      for (int i = 1; i <= 3; i++) {
        if (sm[3 - i] > sm[3] || sm[3] < sm[3 + i]) largest = false;  // :282
      }
      peak = largest ? (1u << 25) : 0u;
    }
    // ---- emit: ascending (intensity, range), unused slots 0 ----
    uint32_t* out = slots + g * (long long)k;
    if (kept) out[kk - 1 - rank] = key | peak;
    if (lane >= kk && lane < k) out[lane] = 0u;
    wave_lds_fence();
  }
}


// ---- round 6: the same filter with the phases after the selection run for TWO rows at once ------------------------------------------------------
// Per row (S-world, profiles/r05_k1_phases.txt) the kernel above issues 35 vector instructions for the load, 115 for the threshold search and the
// masks, and 148 for what follows - candidates to lanes (80), ranking (27), suppression and emission (41) - wave-wide operations on the ~14 candidates
// of a row in a 64-lane wave. Here a wave selects two consecutive rows (both rows' loads issued up front), and when both leave at most 32 candidates
// the 148 run once for the pair: row A's candidates in lanes 0..31, row B's in lanes 32..63 - one packed prefix scan, the owner scan stopped at
// the half (the two row broadcasts become one), the ranking loop over the longer of the two lists, one suppression pass. The price is a second LDS
// window per wave (both rows' bytes must stay readable until the suppression): 38 KB per workgroup, four workgroups per unit instead of seven - which
// is why both rows' loads go out before either is looked at. Rows that do not fit (more than 32 candidates, k > 32, a kept point within three bins
// of a row end) take the one-row phases, written here once more as a lambda. Same results bit for bit (tests/test_kstrongest_gpu.py runs both).
// Selected with CFEAR_K1_PAIR=1 (read once); measured against the kernel above in profiles/r06_k1_pair.txt.
template <int NCH>
__global__ __launch_bounds__(256, 4) void kstrongest_pair_kernel(const uint8_t* __restrict__ polar, uint32_t* __restrict__ slots, int A, int R,
                                                                 long long n_rows, int u_zmin, int k, long long alloc_bytes, int rows_per_wave) {
  constexpr int SUMBITS = NCH == 4 ? 7 : (NCH == 8 ? 8 : 9);
  __shared__ uint4 lds_win[4][2][NCH * 64];
  __shared__ __attribute__((aligned(16))) uint32_t lds_keys[4][64];
  __shared__ uint32_t lds_ge[20];
  __shared__ uint32_t lds_cmp[4][(1 + NCH) * 64];  // per wave: slot offsets (both rows, 16 bits each) and the mask words of row A, then of row B
  __shared__ uint8_t lds_nth[16 * 4];  // lds_nth[n * 4 + r] = position of the r-th set bit of nibble n (a nibble table: with the byte table of the kernel above
                                       // the workgroup is 80 bytes over a quarter of the unit's LDS)
  if (threadIdx.x <= 16) lds_ge[threadIdx.x] = chunk_range_mask((int)threadIdx.x, 16);
  if (threadIdx.x < 16) {
    const uint32_t b = threadIdx.x;
    int r = 0;
    for (int p = 0; p < 4; p++)
      if ((b >> p) & 1u) lds_nth[b * 4 + r++] = (uint8_t)p;
    for (; r < 4; r++) lds_nth[b * 4 + r] = 0;
  }
  __syncthreads();
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* const win0 = reinterpret_cast<uint8_t*>(&lds_win[wave][0][0]);
  uint8_t* const win1 = reinterpret_cast<uint8_t*>(&lds_win[wave][1][0]);
  uint32_t* const keys = &lds_keys[wave][0];
  uint32_t* const cmp_ex = &lds_cmp[wave][0];
  uint32_t* const cmp_w = &lds_cmp[wave][64];  // [NCH][64]: words 0 .. NCH/2-1 row A, NCH/2 .. NCH-1 row B
  const long long scan_bytes = (long long)A * (long long)R;
  const int Tfloor = u_zmin > 1 ? u_zmin : 1;
  const long long g0 = ((long long)blockIdx.x * 4 + wave) * rows_per_wave;
  long long g1 = g0 + rows_per_wave;
  if (g1 > n_rows) g1 = n_rows;
  int Tprev = Tfloor;

  struct Row { long long g, scan; int bearing, head, c_lo, c_hi; bool edge_row; const uint8_t* wp; long long wstart_off; };
  struct Sel { uint32_t w[NCH / 2]; int cnt, lo; };
  auto row_of = [&](long long g) -> Row {
    Row r;
    r.g = g; r.scan = g / A; r.bearing = (int)(g - r.scan * A);
    const long long row_off = g * (long long)R;
    r.wstart_off = (row_off - 6) & ~15LL;
    r.head = (int)(row_off - r.wstart_off);
    r.c_lo = r.wstart_off >= 0 ? 0 : (int)((-r.wstart_off + 15) >> 4);
    r.c_hi = (r.head + R + 6 + 15) >> 4;
    const long long lim = (alloc_bytes - r.wstart_off + 15) >> 4;
    if (lim < r.c_hi) r.c_hi = (int)lim;
    r.edge_row = r.bearing == 0 || r.bearing == A - 1;
    r.wp = polar + r.wstart_off;
    return r;
  };
  // the row's chunks on their way (clamped addresses: no branch around the loads)
  auto issue = [&](const Row& r, uint4 (&v)[NCH]) {
    if (!r.edge_row && r.c_lo == 0) {
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int cc = min(j * 64 + lane, r.c_hi - 1);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r.wp + (uint32_t)(16 * cc)));
        v[j] = make_uint4(t.x, t.y, t.z, t.w);
      }
    } else {
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int cc = min(max(j * 64 + lane, r.c_lo), r.c_hi - 1);
        v[j] = *reinterpret_cast<const uint4*>(r.wp + (uint32_t)(16 * cc));
      }
    }
  };
  // ... into the row's LDS window (scan-masked on the first / last row of an image), then the threshold search of the kernel above -> packed masks
  auto select = [&](const Row& r, uint4 (&v)[NCH], uint8_t* win) -> Sel {
    if (!r.edge_row && r.c_lo == 0) {
#pragma unroll
      for (int j = 0; j < NCH; j++) reinterpret_cast<uint4*>(win)[j * 64 + lane] = v[j];
    } else {
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int c = j * 64 + lane;
        if (c < r.c_lo || c >= r.c_hi) v[j] = make_uint4(0, 0, 0, 0);
        uint4 staged = v[j];
        if (r.edge_row) {
          const long long ca = r.wstart_off + 16LL * c;
          const long long slo = r.scan * scan_bytes - ca, shi = (r.scan + 1) * scan_bytes - ca;
          if (slo > 0 || shi < 16) staged = chunk_keep(staged, (int)(slo > 16 ? 16 : slo), (int)(shi < 0 ? 0 : (shi > 16 ? 16 : shi)));
        }
        reinterpret_cast<uint4*>(win)[c] = staged;
      }
    }
    const int head = r.head;
    const int jt = (head + R - 1) >> 10;
    int hb = head - 16 * lane, tb = head + R - 16 * (jt * 64 + lane);
    hb = hb < 0 ? 0 : (hb > 16 ? 16 : hb);
    tb = tb < 0 ? 0 : (tb > 16 ? 16 : tb);
    const uint32_t vhead = lds_ge[hb];
    const uint32_t vtail = (~lds_ge[tb] & 0x0F0F0F0Fu) & (jt == 0 ? vhead : 0x0F0F0F0Fu);
    const uint32_t vtail2 = 0u;
    wave_lds_fence();
    int lo = Tprev;
    uint32_t m[NCH];
    int cnt = wave_sum_small<SUMBITS>(count_mask<NCH>(v, lo, m, vhead, jt, vtail, vtail2));
    if (cnt < k && lo > Tfloor) {
      lo = Tfloor;
      int lm = lane_max_byte_regs<NCH>(v, vhead, jt, vtail);
      if (lm < Tfloor) lm = 0;
      int tl = 0, th = 256;
      while (th - tl > 1) {
        const int mid = (tl + th) >> 1;
        if (__popcll(__ballot(lm >= mid)) >= k) tl = mid; else th = mid;
      }
      if (tl > lo) lo = tl;
      cnt = wave_sum_small<SUMBITS>(count_mask<NCH>(v, lo, m, vhead, jt, vtail, vtail2));
    }
    if (cnt > 64 && lo < 255) {
      int lm = lane_max_byte_regs<NCH>(v, vhead, jt, vtail);
      int tl = lo, th = 256;
      while (th - tl > 1) {
        const int mid = (tl + th) >> 1;
        if (__popcll(__ballot(lm >= mid)) >= k) tl = mid; else th = mid;
      }
      if (tl > lo) {
        lo = tl;
        cnt = wave_sum_small<SUMBITS>(count_mask<NCH>(v, lo, m, vhead, jt, vtail, vtail2));
      }
    }
    if (cnt > 64) {
      int hi = 256;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        uint32_t mm[NCH];
        const int c2 = wave_sum_small<SUMBITS>(count_mask<NCH>(v, mid, mm, vhead, jt, vtail, vtail2));
        if (c2 >= k) {
          lo = mid; cnt = c2;
#pragma unroll
          for (int j = 0; j < NCH; j++) m[j] = mm[j];
          if (c2 <= 64) break;
        } else {
          hi = mid;
        }
      }
    }
    // (the next row starts one higher already when this one collected more than 24: the pair phases take rows of at most 32)
    Tprev = (cnt > 24 && lo < 255) ? lo + 1 : lo;
    if (cnt > 64) {
      uint32_t mg[NCH];
      const int c_gt = wave_sum_small<SUMBITS>(count_mask<NCH>(v, lo + 1, mg, vhead, jt, vtail, vtail2));
      const int pstar = tie_position(win, head, R, lo, k - c_gt, lane);
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int bp = 16 * (j * 64 + lane) - head;
        const int nb = pstar - bp;
        const uint32_t keep = lds_ge[nb < 0 ? 0 : (nb > 16 ? 16 : nb)];
        m[j] = mg[j] | (m[j] & ~mg[j] & keep);
      }
      cnt = k;
    } else if (u_zmin == 0 && cnt < k && R > cnt && lo == 1) {
      int need = k - cnt;
      if (need > R - cnt) need = R - cnt;
      const int pstar = tie_position(win, head, R, 0, need, lane);
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int bp = 16 * (j * 64 + lane) - head;
        const uint32_t keep = chunk_range_mask(pstar - bp, R - bp);
        m[j] |= (~m[j]) & 0x0F0F0F0Fu & keep;
      }
      cnt += need;
    }
    Sel s;
    pack_masks<NCH>(m, s.w);
    s.cnt = cnt; s.lo = lo;
    return s;
  };
  // set bit number r of a lane's packed mask words -> window offset of that byte
  auto bit_to_woff = [&](const uint32_t* words /* LDS: stride 64 */, int l, int r) -> uint32_t {
    uint32_t sel = 0u;
    int wi = 0;
    bool found = false;
#pragma unroll
    for (int i = 0; i < NCH / 2; i++) {
      const uint32_t wv = words[i * 64 + l];
      const int ci = __popc(wv);
      if (!found) {
        if (r < ci) { sel = wv; wi = i; found = true; } else r -= ci;
      }
    }
    int t = 0;
    const uint32_t lo16 = sel & 0xFFFFu;
    const int c16 = __popc(lo16);
    if (r >= c16) { r -= c16; t = 16; sel >>= 16; } else sel = lo16;
    const uint32_t lo8 = sel & 0xFFu;
    const int c8 = __popc(lo8);
    if (r >= c8) { r -= c8; t += 8; sel >>= 8; } else sel = lo8;
    const uint32_t lo4 = sel & 0xFu;
    const int c4 = __popc(lo4);
    if (r >= c4) { r -= c4; t += 4; sel >>= 4; } else sel = lo4;
    t += lds_nth[(sel & 0xFu) * 4 + r];
    const int j = 2 * wi + ((t >> 2) & 1);
    const int bi = 4 * (t & 3) + (t >> 3);
    return (uint32_t)(16 * (j * 64 + l) + bi);
  };
  // 13-tap non-max suppression of one kept point (radar_filters.cpp:238-298) with all seven scores present (or masked by `covered`)
  auto peak_of = [&](const uint8_t* win, int head, int mpos, bool mask_scores, uint32_t covered) -> uint32_t {
    const int off0 = head + mpos - 6;
    int bv[13];
#pragma unroll
    for (int t = 0; t < 13; t++) bv[t] = win[off0 + t];
    int sm[7];
    sm[0] = ((bv[0] + bv[1]) + (bv[2] + bv[3])) + ((bv[4] + bv[5]) + bv[6]);
#pragma unroll
    for (int u = 1; u < 7; u++) sm[u] = sm[u - 1] - bv[u - 1] + bv[u + 6];
    if (mask_scores) {
#pragma unroll
      for (int u = 0; u < 7; u++)
        if (!((covered >> u) & 1u)) sm[u] = 0;
    }
    bool largest = true;
#pragma unroll
    for (int i = 1; i <= 3; i++)
      if (sm[3 - i] > sm[3] || sm[3] < sm[3 + i]) largest = false;
    return largest ? (1u << 25) : 0u;
  };
  // the one-row phases of the kernel above (candidates to lanes, ranking, suppression, emission) from a row's packed masks
  auto finish_one = [&](const Row& r, const Sel& s, uint8_t* win) {
    const int head = r.head, cnt = s.cnt;
    const int C = cnt < 64 ? cnt : 64;
    const int kk = k < C ? k : C;
    uint32_t key = 0u;
    {
      int c = 0;
#pragma unroll
      for (int i = 0; i < NCH / 2; i++) c += __popc(s.w[i]);
      const int ex = wave_inclusive_scan(c) - c;
      keys[lane] = 0u;
      cmp_ex[lane] = (uint32_t)ex;
#pragma unroll
      for (int i = 0; i < NCH / 2; i++) cmp_w[i * 64 + lane] = s.w[i];
      wave_lds_fence();
      if (c > 0 && ex < 64) keys[ex] = (uint32_t)(lane + 1);
      wave_lds_fence();
      const int owner = wave_inclusive_max((int)keys[lane]) - 1;
      if (lane < C) {
        const uint32_t woff = bit_to_woff(cmp_w, owner, lane - (int)cmp_ex[owner]);
        key = (woff - (uint32_t)head) | ((uint32_t)win[woff] << 16) | (1u << 24);
      }
    }
    bool kept;
    int rank = 0;
    wave_lds_fence();
    if (C > 2 * k) {
      const uint32_t k24 = key & 0xFFFFFFu;
      uint32_t klo = (uint32_t)s.lo << 16, khi = 256u << 16;
      while (khi - klo > 1u) {
        const uint32_t mid = (klo + khi) >> 1;
        const int c = __popcll(__ballot(k24 >= mid));
        if (c >= k) { klo = mid; if (c == k) break; } else khi = mid;
      }
      kept = k24 >= klo;
      const unsigned long long keptb = __ballot(kept);
      const int kpos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(keptb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)keptb, 0u));
      if (kept) keys[kpos] = key;
      if (lane >= kk && lane < kk + 4) keys[lane] = 0u;
      wave_lds_fence();
      for (int j = 0; j < kk; j += 4) {
        const uint4 ka = reinterpret_cast<const uint4*>(keys)[j >> 2];
        rank += (ka.x > key) ? 1 : 0; rank += (ka.y > key) ? 1 : 0; rank += (ka.z > key) ? 1 : 0; rank += (ka.w > key) ? 1 : 0;
      }
    } else {
      keys[lane] = key;
      wave_lds_fence();
      for (int j = 0; j < C; j += 4) {
        const uint4 ka = reinterpret_cast<const uint4*>(keys)[j >> 2];
        rank += (ka.x > key) ? 1 : 0; rank += (ka.y > key) ? 1 : 0; rank += (ka.z > key) ? 1 : 0; rank += (ka.w > key) ? 1 : 0;
      }
      kept = lane < C && rank < kk;
    }
    const int mpos = (int)(key & 0xFFFFu);
    uint32_t peak = 0;
    const bool interior = mpos >= 3 && mpos < R - 3;
    uint32_t covered = 0x7Fu;
    const bool edge_points = __ballot(kept && !interior) != 0;
    if (edge_points) {
      if (R >= 24) {
        const bool ik = kept && interior;
        int marks = (ik && mpos <= 8) ? (1 << (mpos - 3)) : 0;
        marks |= (ik && mpos >= R - 9) ? (1 << (6 + mpos - (R - 9))) : 0;
        marks = __builtin_amdgcn_readlane(wave_inclusive_or(marks), 63);
        uint32_t es = (uint32_t)marks & 0x3Fu, ee = ((uint32_t)marks >> 6) & 0x3Fu;
        es |= es << 1; es |= es << 2; es |= es << 3;
        ee |= ee << 1; ee |= ee << 2; ee |= ee << 3;
        const uint32_t c_lo = ((es << 3) >> (mpos < 3 ? mpos : 0)) & 0x7Fu;
        const uint32_t c_hi = (ee >> (mpos >= R - 3 ? mpos - R + 9 : 0)) & 0x7Fu;
        covered = interior ? 0x7Fu : (mpos < 3 ? c_lo : c_hi);
      } else {
        covered = 0;
        unsigned long long kb = __ballot(kept);
        while (kb) {
          const int i = __ffsll((long long)kb) - 1;
          kb &= kb - 1;
          const int mi = (int)(__builtin_amdgcn_readlane((int)key, i) & 0xFFFF);
          if (mi >= 3 && mi < R - 3) {
#pragma unroll
            for (int t = 0; t < 7; t++) {
              const int rr = mpos - 3 + t;
              if (rr >= mi - 3 && rr <= mi + 3) covered |= 1u << t;
            }
          }
        }
      }
    }
    if (kept) peak = peak_of(win, head, mpos, edge_points, covered);
    uint32_t* out = slots + r.g * (long long)k;
    if (kept) out[kk - 1 - rank] = key | peak;
    if (lane >= kk && lane < k) out[lane] = 0u;
    wave_lds_fence();
  };
  // ... and for two rows of at most 32 candidates each at once: row A in lanes 0..31, row B in lanes 32..63. Returns false (nothing written) when a
  // kept point sits within three bins of a row end: the caller runs the one-row phases then.
  auto finish_two = [&](const Row& ra, const Sel& sa, const Row& rb, const Sel& sb) -> bool {
    const int half = lane >> 5, slot = lane & 31;
    const int C = half ? sb.cnt : sa.cnt, kk = k < C ? k : C, head = half ? rb.head : ra.head;
    const uint8_t* win = half ? win1 : win0;
    uint32_t key = 0u;
    {
      int ca = 0, cb = 0;
#pragma unroll
      for (int i = 0; i < NCH / 2; i++) { ca += __popc(sa.w[i]); cb += __popc(sb.w[i]); }
      const int c2 = ca | (cb << 16);
      const int ex2 = wave_inclusive_scan(c2) - c2;  // (both prefix sums in one scan: each stays below 2^16)
      keys[lane] = 0u;
      cmp_ex[lane] = (uint32_t)ex2;
#pragma unroll
      for (int i = 0; i < NCH / 2; i++) { cmp_w[i * 64 + lane] = sa.w[i]; cmp_w[(NCH / 2 + i) * 64 + lane] = sb.w[i]; }
      wave_lds_fence();
      if (ca > 0) keys[ex2 & 0xFFFF] = (uint32_t)(lane + 1);        // < 32: the row has at most 32 candidates
      if (cb > 0) keys[32 + (ex2 >> 16)] = (uint32_t)(lane + 1);
      wave_lds_fence();
      int o = (int)keys[lane];  // owner scan inside the halves: the row steps and the broadcast into rows 1 and 3, not the one across the middle
      o = dpp_max_i<0x111, 0xF>(o); o = dpp_max_i<0x112, 0xF>(o); o = dpp_max_i<0x114, 0xF>(o); o = dpp_max_i<0x118, 0xF>(o);
      o = dpp_max_i<0x142, 0xA>(o);
      const int owner = o - 1;
      if (slot < C) {
        const uint32_t e2 = cmp_ex[owner];
        const uint32_t woff = bit_to_woff(cmp_w + (half ? (NCH / 2) * 64 : 0), owner, slot - (int)(half ? (e2 >> 16) : (e2 & 0xFFFFu)));
        key = (woff - (uint32_t)head) | ((uint32_t)win[woff] << 16) | (1u << 24);
      }
    }
    int rank = 0;
    wave_lds_fence();
    keys[lane] = key;  // lanes past a row's candidates hold key 0: smaller than any candidate
    wave_lds_fence();
    const int Cmax = sa.cnt > sb.cnt ? sa.cnt : sb.cnt;
    for (int j = 0; j < Cmax; j += 4) {
      const uint4 ka = reinterpret_cast<const uint4*>(keys)[(32 * half + j) >> 2];
      rank += (ka.x > key) ? 1 : 0; rank += (ka.y > key) ? 1 : 0; rank += (ka.z > key) ? 1 : 0; rank += (ka.w > key) ? 1 : 0;
    }
    const bool kept = slot < C && rank < kk;
    const int mpos = (int)(key & 0xFFFFu);
    const bool interior = mpos >= 3 && mpos < R - 3;
    if (__ballot(kept && !interior) != 0) { wave_lds_fence(); return false; }
    const uint32_t peak = kept ? peak_of(win, head, mpos, false, 0x7Fu) : 0u;
    uint32_t* out = slots + (half ? rb.g : ra.g) * (long long)k;
    if (kept) out[kk - 1 - rank] = key | peak;
    if (slot >= kk && slot < k) out[slot] = 0u;
    wave_lds_fence();
    return true;
  };

  // (Asking for the next pair's rows as soon as the registers are free - before the pair phases - was measured too: the loads have to be unconditional
  // for the wait counters to stay countable, the rows past a wave's last are then read again, and at six rows per wave that is a third more traffic:
  // 461 us against 417 us per 1536 scans, profiles/r06_k1_pair.txt.)
  for (long long g = g0; g < g1; g += 2) {
    const bool two = g + 1 < g1;
    const Row ra = row_of(g), rb = row_of(two ? g + 1 : g);
    uint4 va[NCH], vb[NCH];
    issue(ra, va);
    if (two) issue(rb, vb);
    const Sel sa = select(ra, va, win0);
    if (!two) { finish_one(ra, sa, win0); break; }
    const Sel sb = select(rb, vb, win1);
    if (sa.cnt <= 32 && sb.cnt <= 32 && k <= 32 && finish_two(ra, sa, rb, sb)) continue;
    finish_one(ra, sa, win0);
    finish_one(rb, sb, win1);
  }
}
}  // namespace

// Launch-shape knobs live in the context (cfear_tune, include/cfear_hip.h): occupancy variant and rows per wave.
int cfear_launch_kstrongest(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, uint32_t* d_slots, hipStream_t stream) {
  const int A = ctx->A, R = ctx->R, k = ctx->par.k_strongest;
  if (!d_polar || !d_slots || n_scans <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "kstrongest: null buffer or n_scans <= 0");
  if (k < 1 || k > 64) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "kstrongest: k_strongest must be in 1..64");
  if ((reinterpret_cast<uintptr_t>(d_polar) & 15) != 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "kstrongest: polar buffer must be 16-byte aligned");
  const long long n_rows = (long long)n_scans * A;
  const long long alloc = n_rows * R;
  const int u_zmin = (int)(uint8_t)(int)ctx->par.z_min;  // float -> int (radar_filters.cpp:198) -> uchar (:212)
  // one resident wave per SIMD slot (256 CUs x 4 SIMDs x occupancy); each wave walks consecutive rows
  const int occ_eff = (R + 27 <= 4 * 1024) ? (ctx->tune_k1_occ >= 7 ? 7 : (ctx->tune_k1_occ <= 5 ? 5 : 6)) : (R + 27 <= 8 * 1024 ? 3 : 2);
  // A wave walks a few consecutive rows (the threshold of one azimuth is the first guess for the next): four rows
  // per wave measured best from 256-scan to 1024-scan launches (shorter: every row pays the cold threshold search;
  // longer: fewer, longer workgroups balance worse), six from 1536 scans up (round 3, inside the bench's timed region at 4608
  // scans: 1041 -> 1014 us, 0.754 -> 0.774 of the HBM peak; 8 and 12 the same, 16 worse at 1536). Small launches spread their
  // rows over the resident slots.
  const long long slots_total = 1024LL * occ_eff;
  int rows_per_wave = (int)((n_rows + slots_total - 1) / slots_total);
  const int rows_cap = ctx->tune_k1_rows > 0 ? ctx->tune_k1_rows : (n_scans >= 1536 ? 6 : 4);
  if (rows_per_wave > rows_cap) rows_per_wave = rows_cap;
  if (rows_per_wave < 1) rows_per_wave = 1;
  static const bool pair = getenv("CFEAR_K1_PAIR") != nullptr && atoi(getenv("CFEAR_K1_PAIR")) != 0;  // the two-rows-at-once variant (A/B: tools/gpu_time_k1_pair.sh)
  if (pair && rows_per_wave < 2) rows_per_wave = 2;  // (small launches too: a wave of the variant wants a pair)
  const long long n_waves = (n_rows + rows_per_wave - 1) / rows_per_wave;
  const long long blocks = (n_waves + 3) / 4;
  dim3 grid((unsigned)blocks), block(256);
  const int occ = ctx->tune_k1_occ;
  if (pair && R + 27 <= 4 * 1024) {
    hipLaunchKernelGGL((kstrongest_pair_kernel<4>), grid, block, 0, stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc, rows_per_wave);
    CFEAR_HIP_CHECK(ctx, hipGetLastError());
    return CFEAR_OK;
  }
  if (R + 27 <= 4 * 1024) {
    if (occ >= 7)
      hipLaunchKernelGGL((kstrongest_kernel<4, 7>), grid, block, 0, stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc, rows_per_wave);
    else if (occ <= 5)
      hipLaunchKernelGGL((kstrongest_kernel<4, 5>), grid, block, 0, stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc, rows_per_wave);
    else
      hipLaunchKernelGGL((kstrongest_kernel<4, 6>), grid, block, 0, stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc, rows_per_wave);
  } else if (R + 27 <= 8 * 1024)
    hipLaunchKernelGGL((kstrongest_kernel<8, 3>), grid, block, 0, stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc, rows_per_wave);
  else if (R + 27 <= 16 * 1024)
    hipLaunchKernelGGL((kstrongest_kernel<16, 2>), grid, block, 0, stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc, rows_per_wave);
  else
    return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "kstrongest: R > 16357 range bins not supported");
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}
