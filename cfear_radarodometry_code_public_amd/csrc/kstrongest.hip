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
//   1. bounds the k-th largest intensity from below by the k-th largest per-lane maximum,
//   2. counts bytes >= T with SWAR compares (4 bytes / 5 VALU ops) and narrows T by bisection
//      until k <= count <= 64 (usually the first probe),
//   3. compacts the <= 64 candidates through LDS, ranks them (key includes the range, so ties go
//      to the larger range bin exactly like std::pair<uchar,int> ordering) and
//   4. handles rows with > 64 ties at the threshold intensity by a backwards positional scan.
// No block-level barrier is used: the four waves of a block are independent.
#include "common.h"

namespace {

constexpr uint32_t HI = 0x80808080u;

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// wave-wide sum of a small non-negative per-lane integer (< 2^BITS) by ballot bit-slicing
template <int BITS>
__device__ __forceinline__ int wave_sum_small(int v) {
  int total = 0;
#pragma unroll
  for (int b = 0; b < BITS; b++) total += __popcll(__ballot((v >> b) & 1)) << b;
  return total;
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

// per-byte (x >= T) as 0x80 flags; brep = (T & 0x7f) replicated, THIGH = (T >= 128)
template <bool THIGH>
__device__ __forceinline__ uint32_t ge_flags(uint32_t x, uint32_t brep) {
  const uint32_t d = (x | HI) - brep;
  return (THIGH ? (x & d) : (x | d)) & HI;
}

// Chunk candidate mask layout: bit (8*b + d) <-> byte b of dword d (byte index 4*d + b in the chunk).
template <int NCH, bool THIGH>
__device__ __forceinline__ int count_mask_t(const uint4 (&v)[NCH], uint32_t brep, uint32_t (&m)[NCH]) {
  int cnt = 0;
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    const uint32_t g0 = ge_flags<THIGH>(v[j].x, brep), g1 = ge_flags<THIGH>(v[j].y, brep);
    const uint32_t g2 = ge_flags<THIGH>(v[j].z, brep), g3 = ge_flags<THIGH>(v[j].w, brep);
    cnt += __popc(g0) + __popc(g1) + __popc(g2) + __popc(g3);
    m[j] = (g0 >> 7) | (g1 >> 6) | (g2 >> 5) | (g3 >> 4);
  }
  return cnt;
}
// per-lane count of bytes >= T (1 <= T <= 255) and their masks; T == 256 -> none
template <int NCH>
__device__ __forceinline__ int count_mask(const uint4 (&v)[NCH], int T, uint32_t (&m)[NCH]) {
  if (T >= 256) {
#pragma unroll
    for (int j = 0; j < NCH; j++) m[j] = 0;
    return 0;
  }
  const uint32_t brep = (uint32_t)(T & 0x7f) * 0x01010101u;
  return (T >= 128) ? count_mask_t<NCH, true>(v, brep, m) : count_mask_t<NCH, false>(v, brep, m);
}

// per-lane maximum byte of the chunks (v_pk_max_u16 on the odd bytes and on the even bytes << 8)
template <int NCH>
__device__ __forceinline__ int lane_max_byte(const uint4 (&v)[NCH]) {
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  us2 ao = {0, 0}, ae = {0, 0};
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    const uint32_t w[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const uint32_t o = w[d], e = w[d] << 8;
      us2 vo = __builtin_bit_cast(us2, o), ve = __builtin_bit_cast(us2, e);
      ao = __builtin_elementwise_max(ao, vo);
      ae = __builtin_elementwise_max(ae, ve);
    }
  }
  const int m0 = ao.x >> 8, m1 = ao.y >> 8, m2 = ae.x >> 8, m3 = ae.y >> 8;
  const int a = m0 > m1 ? m0 : m1, b = m2 > m3 ? m2 : m3;
  return a > b ? a : b;
}

template <int NCH>
__global__ __launch_bounds__(256) void kstrongest_kernel(const uint8_t* __restrict__ polar,
                                                         uint32_t* __restrict__ slots, int A, int R,
                                                         long long n_rows, int u_zmin, int k,
                                                         long long alloc_bytes) {
  constexpr int WIN = NCH * 1024;  // LDS window bytes per wave
  __shared__ uint4 lds_win[4][NCH * 64];
  __shared__ uint32_t lds_keys[4][64];
  const int lane = lane_id();
  const int wave = threadIdx.x >> 6;
  uint8_t* const win = reinterpret_cast<uint8_t*>(&lds_win[wave][0]);
  uint32_t* const keys = &lds_keys[wave][0];
  const uintptr_t base = reinterpret_cast<uintptr_t>(polar);
  const uintptr_t alloc_end = base + (uintptr_t)alloc_bytes;
  const long long scan_bytes = (long long)A * (long long)R;
  const int Tfloor = u_zmin > 1 ? u_zmin : 1;

  for (long long g = (long long)blockIdx.x * 4 + wave; g < n_rows; g += (long long)gridDim.x * 4) {
    const long long scan = g / A;
    const int bearing = (int)(g - scan * A);
    const uintptr_t row_addr = base + (uintptr_t)(g * (long long)R);
    const uintptr_t scan_lo = base + (uintptr_t)(scan * scan_bytes), scan_hi = scan_lo + (uintptr_t)scan_bytes;
    const uintptr_t wstart = (row_addr - 6) & ~(uintptr_t)15;
    const int head = (int)(row_addr - wstart);  // 6..21: window offset of range bin 0
    const int need_bytes = head + R + 6;

    // ---- load: HBM -> VGPR (row-masked) and LDS (scan-masked, keeps the cross-row halo) ----
    uint4 v[NCH];
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      const int c = j * 64 + lane;
      const uintptr_t ca = wstart + (uintptr_t)(16 * c);
      uint4 raw = make_uint4(0, 0, 0, 0);
      if (16 * c < need_bytes && ca >= base && ca < alloc_end) raw = *reinterpret_cast<const uint4*>(ca);
      // bytes outside this scan's image read as 0 (the reference's unchecked cv::Mat::at would run
      // off the buffer there, radar_filters.cpp:260)
      const long long slo = (long long)scan_lo - (long long)ca, shi = (long long)scan_hi - (long long)ca;
      uint4 staged = raw;
      if (slo > 0 || shi < 16) staged = chunk_keep(raw, (int)(slo > 16 ? 16 : slo), (int)(shi < 0 ? 0 : (shi > 16 ? 16 : shi)));
      reinterpret_cast<uint4*>(win)[c] = staged;
      const int rlo = head - 16 * c, rhi = head + R - 16 * c;
      v[j] = (rlo > 0 || rhi < 16) ? chunk_keep(raw, rlo > 16 ? 16 : rlo, rhi < 0 ? 0 : rhi) : raw;
    }

    wave_lds_fence();

    // ---- threshold search ----
    int lo = Tfloor;
    if (k <= 64) {
      int lm = lane_max_byte<NCH>(v);
      if (lm < Tfloor) lm = 0;
      // k-th largest lane maximum: largest t with #{lanes: lm >= t} >= k (t = 0 always qualifies)
      int tl = 0, th = 256;
      while (th - tl > 1) {
        const int mid = (tl + th) >> 1;
        if (__popcll(__ballot(lm >= mid)) >= k) tl = mid; else th = mid;
      }
      if (tl > lo) lo = tl;  // count(bytes >= tl) >= k is guaranteed
    }
    uint32_t m[NCH];
    int ln = count_mask<NCH>(v, lo, m);
    int cnt = wave_sum_small<(NCH == 4 ? 7 : (NCH == 8 ? 8 : 9))>(ln);
    if (cnt > 64) {
      int hi = 256;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        uint32_t mm[NCH];
        const int l2 = count_mask<NCH>(v, mid, mm);
        const int c2 = wave_sum_small<(NCH == 4 ? 7 : (NCH == 8 ? 8 : 9))>(l2);
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
    // ---- ties: > 64 bytes equal to the threshold intensity, or z_min == 0 and zeros are needed ----
    int tie_val = -1, c_gt = 0;
    if (cnt > 64) {
      tie_val = lo;
      uint32_t mg[NCH];
      const int l2 = count_mask<NCH>(v, lo + 1, mg);
      c_gt = wave_sum_small<(NCH == 4 ? 7 : (NCH == 8 ? 8 : 9))>(l2);
      // m (>= lo) becomes the "== lo" mask, mg the "> lo" mask
#pragma unroll
      for (int j = 0; j < NCH; j++) { m[j] &= ~mg[j]; uint32_t t = m[j]; m[j] = mg[j]; mg[j] = t; }
      // now m = gt mask, mg = eq mask
      int need = k - c_gt;
      // backwards positional scan over the LDS row for the need-th largest position with byte == lo
      int pstar = 0, acc = 0;
      for (int tb = R - 64; tb > -64; tb -= 64) {
        const int pos = tb + lane;
        const bool hit = pos >= 0 && win[head + pos] == (uint8_t)tie_val;
        const unsigned long long b = __ballot(hit);
        const int c = __popcll(b);
        if (acc + c >= need) {
          unsigned long long bb = b;
          int want = need - acc;  // want-th highest set bit
          int bit = 63;
          while (true) {
            bit = 63 - __clzll(bb);
            if (--want == 0) break;
            bb &= ~(1ull << bit);
          }
          pstar = tb + bit;
          break;
        }
        acc += c;
      }
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int bp = 16 * (j * 64 + lane) - head;  // range bin of byte 0 of this chunk
        uint32_t keep = 0;
        for (int bi = 0; bi < 16; bi++) {
          const int pos = bp + bi;
          if (pos >= pstar && pos < R) keep |= 1u << (8 * (bi & 3) + (bi >> 2));
        }
        m[j] |= mg[j] & keep;
      }
      cnt = k;
    } else if (u_zmin == 0 && cnt < k && R > cnt && lo == 1) {
      // z_min == 0: zero-valued bins are candidates too; take the largest ranges among them
      int need = k - cnt;
      if (need > R - cnt) need = R - cnt;
      int pstar = 0, acc = 0;
      for (int tb = R - 64; tb > -64; tb -= 64) {
        const int pos = tb + lane;
        const bool hit = pos >= 0 && win[head + pos] == 0;
        const unsigned long long b = __ballot(hit);
        const int c = __popcll(b);
        if (acc + c >= need) {
          unsigned long long bb = b;
          int want = need - acc, bit = 63;
          while (true) {
            bit = 63 - __clzll(bb);
            if (--want == 0) break;
            bb &= ~(1ull << bit);
          }
          pstar = tb + bit;
          break;
        }
        acc += c;
      }
#pragma unroll
      for (int j = 0; j < NCH; j++) {
        const int bp = 16 * (j * 64 + lane) - head;
        uint32_t keep = 0;
        for (int bi = 0; bi < 16; bi++) {
          const int pos = bp + bi;
          if (pos >= pstar && pos < R) keep |= 1u << (8 * (bi & 3) + (bi >> 2));
        }
        m[j] |= (~m[j]) & 0x0F0F0F0Fu & keep;
      }
      cnt += need;
    }

    // ---- compaction of the <= 64 candidates into LDS keys ----
    wave_lds_fence();
    int nbase = 0;
#pragma unroll
    for (int j = 0; j < NCH; j++) {
      uint32_t mj = m[j];
      while (true) {
        const bool has = mj != 0;
        const unsigned long long b = __ballot(has);
        if (b == 0) break;
        if (has) {
          const int t = __ffs(mj) - 1;
          const int bi = 4 * (t & 7) + (t >> 3);  // byte index inside the chunk
          const int woff = 16 * (j * 64 + lane) + bi;
          const uint32_t inten = win[woff];
          const int pos = woff - head;
          const int slot = nbase + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
          if (slot < 64) keys[slot] = (uint32_t)pos | (inten << 16) | (1u << 24);
          mj &= mj - 1;
        }
        nbase += __popcll(b);
      }
    }
    wave_lds_fence();
    const int C = cnt < 64 ? cnt : 64;
    const int kk = k < C ? k : C;  // number of emitted points
    const uint32_t key = lane < C ? keys[lane] : 0u;
    int rank = 0;
    for (int j = 0; j < C; j++) rank += (keys[j] > key) ? 1 : 0;
    const bool kept = lane < C && rank < kk;
    const int mpos = (int)(key & 0xFFFFu);

    // ---- axial non-max suppression (radar_filters.cpp:238-298) on the kept points ----
    uint32_t peak = 0;
    const bool interior = mpos >= 3 && mpos < R - 3;  // :251
    uint32_t covered = 0x7Fu;
    if (__ballot(kept && !interior) != 0) {
      // scores exist only where a valid kept point's +-3 window put them in the map (:253-263);
      // everything else reads as the unordered_map default 0 (:271-276)
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
    if (kept) {
      int byt[13];
#pragma unroll
      for (int t = 0; t < 13; t++) {
        int off = head + mpos - 6 + t;
        off = off < 0 ? 0 : (off > WIN - 1 ? WIN - 1 : off);
        byt[t] = win[off];
      }
      int s[7];
#pragma unroll
      for (int t = 0; t < 7; t++) {
        s[t] = byt[t] + byt[t + 1] + byt[t + 2] + byt[t + 3] + byt[t + 4] + byt[t + 5] + byt[t + 6];
        if (!((covered >> t) & 1u)) s[t] = 0;
      }
      bool largest = true;
#pragma unroll
      for (int i = 1; i <= 3; i++) {
        if (s[3 - i] > s[3] || s[3] < s[3 + i]) largest = false;  // :282
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

}  // namespace

int cfear_launch_kstrongest(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, uint32_t* d_slots) {
  const int A = ctx->A, R = ctx->R, k = ctx->par.k_strongest;
  if (!d_polar || !d_slots || n_scans <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "kstrongest: null buffer or n_scans <= 0");
  if (k < 1 || k > 64) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "kstrongest: k_strongest must be in 1..64");
  if ((reinterpret_cast<uintptr_t>(d_polar) & 15) != 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "kstrongest: polar buffer must be 16-byte aligned");
  const long long n_rows = (long long)n_scans * A;
  const long long alloc = n_rows * R;
  const int u_zmin = (int)(uint8_t)(int)ctx->par.z_min;  // float -> int (radar_filters.cpp:198) -> uchar (:212)
  long long blocks = (n_rows + 3) / 4;
  const long long cap = 256LL * 8;
  if (blocks > cap) blocks = cap;
  dim3 grid((unsigned)blocks), block(256);
  if (R + 27 <= 4 * 1024)
    hipLaunchKernelGGL(kstrongest_kernel<4>, grid, block, 0, ctx->stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc);
  else if (R + 27 <= 8 * 1024)
    hipLaunchKernelGGL(kstrongest_kernel<8>, grid, block, 0, ctx->stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc);
  else if (R + 27 <= 16 * 1024)
    hipLaunchKernelGGL(kstrongest_kernel<16>, grid, block, 0, ctx->stream, d_polar, d_slots, A, R, n_rows, u_zmin, k, alloc);
  else
    return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "kstrongest: R > 16357 range bins not supported");
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}
