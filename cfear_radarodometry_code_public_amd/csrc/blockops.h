// blockops.h -- workgroup-level primitives for gfx950 (wave64). Every function here must be
// reached by ALL threads of the block (they contain __syncthreads()).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cfear_dev {

// every scratch array handed to the block primitives below is LDS: an LDS-typed pointer turns the partial-sum
// loops into independent ds_read instructions (through a generic pointer they are dependent flat loads)
#define CFEAR_LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

__device__ __forceinline__ int lane_id() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

// ---- wave reductions (64 lanes) ----
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
// DPP prefix pattern (Hillis-Steele inside the 16-lane rows, then two row broadcasts): after the six steps lane i holds
// op over lanes 0..i, so lane 63 holds the wave result. No LDS crossbar round trips (ds_bpermute) on the way.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_pull(int own, int v) {  // lanes without a source lane get `own`
  return __builtin_amdgcn_update_dpp(own, v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += dpp_pull<0x111, 0xF>(0, v);  // row_shr:1
  v += dpp_pull<0x112, 0xF>(0, v);  // row_shr:2
  v += dpp_pull<0x114, 0xF>(0, v);  // row_shr:4
  v += dpp_pull<0x118, 0xF>(0, v);  // row_shr:8
  v += dpp_pull<0x142, 0xA>(0, v);  // row_bcast15 -> rows 1, 3
  v += dpp_pull<0x143, 0xC>(0, v);  // row_bcast31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ unsigned wave_inclusive_scan_u(unsigned v) {  // the same modulo 2^32 (packed counters that use bit 31)
  v += (unsigned)dpp_pull<0x111, 0xF>(0, (int)v);
  v += (unsigned)dpp_pull<0x112, 0xF>(0, (int)v);
  v += (unsigned)dpp_pull<0x114, 0xF>(0, (int)v);
  v += (unsigned)dpp_pull<0x118, 0xF>(0, (int)v);
  v += (unsigned)dpp_pull<0x142, 0xA>(0, (int)v);
  v += (unsigned)dpp_pull<0x143, 0xC>(0, (int)v);
  return v;
}
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_inclusive_scan(v), 63); }
#define CFEAR_DPP_FLOAT_REDUCE(NAME, OP)                                                          \
  __device__ __forceinline__ float NAME(float v) {                                                \
    int b = __float_as_int(v);                                                                    \
    b = __float_as_int(OP(__int_as_float(b), __int_as_float(dpp_pull<0x111, 0xF>(b, b))));        \
    b = __float_as_int(OP(__int_as_float(b), __int_as_float(dpp_pull<0x112, 0xF>(b, b))));        \
    b = __float_as_int(OP(__int_as_float(b), __int_as_float(dpp_pull<0x114, 0xF>(b, b))));        \
    b = __float_as_int(OP(__int_as_float(b), __int_as_float(dpp_pull<0x118, 0xF>(b, b))));        \
    b = __float_as_int(OP(__int_as_float(b), __int_as_float(dpp_pull<0x142, 0xA>(b, b))));        \
    b = __float_as_int(OP(__int_as_float(b), __int_as_float(dpp_pull<0x143, 0xC>(b, b))));        \
    return __int_as_float(__builtin_amdgcn_readlane(b, 63));                                      \
  }
CFEAR_DPP_FLOAT_REDUCE(wave_min, fminf)
CFEAR_DPP_FLOAT_REDUCE(wave_max, fmaxf)
#undef CFEAR_DPP_FLOAT_REDUCE

// scratch: >= 32 elements of T in LDS. Result broadcast to all threads.
__device__ __forceinline__ float block_min(float v, float* scratch) {
  auto* sl = CFEAR_LDS_PTR(float, scratch);
  v = wave_min(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane_id() == 0) sl[w] = v;
  __syncthreads();
  float r = sl[0];
  for (int i = 1; i < nw; i++) r = fminf(r, sl[i]);
  return r;
}
__device__ __forceinline__ float block_max(float v, float* scratch) {
  auto* sl = CFEAR_LDS_PTR(float, scratch);
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane_id() == 0) sl[w] = v;
  __syncthreads();
  float r = sl[0];
  for (int i = 1; i < nw; i++) r = fmaxf(r, sl[i]);
  return r;
}
// bounding box in one go: v = {min x, max x, min y, max y} per thread -> block-wide values in every thread.
// scratch: >= 4 * 32 floats... uses 4 floats per wave (<= 16 waves -> 64 floats).
template <int NT = 0>
__device__ __forceinline__ void block_bounds(float v[4], float* scratch) {
  auto* sl = CFEAR_LDS_PTR(float, scratch);
  v[0] = wave_min(v[0]); v[1] = wave_max(v[1]); v[2] = wave_min(v[2]); v[3] = wave_max(v[3]);
  const int w = threadIdx.x >> 6, nw = NT ? (NT + 63) >> 6 : (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane_id() == 0) { sl[4 * w] = v[0]; sl[4 * w + 1] = v[1]; sl[4 * w + 2] = v[2]; sl[4 * w + 3] = v[3]; }
  __syncthreads();
  for (int i = 0; i < nw; i++) {
    v[0] = fminf(v[0], sl[4 * i]); v[1] = fmaxf(v[1], sl[4 * i + 1]);
    v[2] = fminf(v[2], sl[4 * i + 2]); v[3] = fmaxf(v[3], sl[4 * i + 3]);
  }
}
__device__ __forceinline__ int block_sum(int v, int* scratch) {
  auto* sl = CFEAR_LDS_PTR(int, scratch);
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane_id() == 0) sl[w] = v;
  __syncthreads();
  int r = 0;
  for (int i = 0; i < nw; i++) r += sl[i];
  return r;
}

// Exclusive prefix sum of one int per thread; *total = block sum. scratch: >= 32 ints in LDS.
// NT: threads of the workgroup when the caller knows them at compile time (blockDim.x is a load from the dispatch packet:
// a round trip to memory in every out-of-line function that asks for it)
template <int NT = 0>
__device__ __forceinline__ int block_exclusive_scan(int v, int* scratch, int* total) {
  auto* sl = CFEAR_LDS_PTR(int, scratch);
  const int lane = lane_id(), w = threadIdx.x >> 6, nw = NT ? (NT + 63) >> 6 : (blockDim.x + 63) >> 6;
  const int inc = wave_inclusive_scan(v);
  __syncthreads();
  if (lane == 63) sl[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < nw; i++) {
    const int s = sl[i];
    if (i < w) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

// The same with ONE barrier: the partial sums go to a 16-int slot of `scratch` (slot 0..3) that no wave may still be reading,
// i.e. between this scan and the previous one that used the same slot there must be a barrier (any barrier) - back-to-back
// scans alternate slots. (The two-barrier version protects its scratch with a barrier of its own; at 8 waves that is ~0.5 us
// per scan, and the feature kernel has seven of them per scan of the radar.)
template <int NT = 0>
__device__ __forceinline__ int block_exclusive_scan_1b(int v, int* scratch, int slot, int* total) {
  auto* sl = CFEAR_LDS_PTR(int, scratch) + 16 * slot;
  const int lane = lane_id(), w = threadIdx.x >> 6, nw = NT ? (NT + 63) >> 6 : (blockDim.x + 63) >> 6;
  const int inc = wave_inclusive_scan(v);
  if (lane == 63) sl[w] = inc;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < nw; i++) {
    const int s = sl[i];
    if (i < w) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

// The one-barrier scan over unsigned values with a block-wide OR of a per-thread flag riding along (entries 8..15 of the slot:
// at most 8 waves): *any = some thread of the block raised its flag.
template <int NT = 0>
__device__ __forceinline__ unsigned block_exclusive_scan_1b_flag(unsigned v, bool flag, int* scratch, int slot, unsigned* total, bool* any) {
  auto* sl = CFEAR_LDS_PTR(int, scratch) + 16 * slot;
  const int lane = lane_id(), w = threadIdx.x >> 6, nw = NT ? (NT + 63) >> 6 : (blockDim.x + 63) >> 6;
  const unsigned inc = wave_inclusive_scan_u(v);
  const bool wf = __ballot(flag) != 0ull;
  if (lane == 63) { sl[w] = (int)inc; sl[8 + w] = wf ? 1 : 0; }
  __syncthreads();
  unsigned base = 0, tot = 0;
  int f = 0;
  for (int i = 0; i < nw; i++) {
    const unsigned s = (unsigned)sl[i];
    if (i < w) base += s;
    tot += s;
    f |= sl[8 + i];
  }
  *total = tot; *any = f != 0;
  return base + inc - v;
}

// Exclusive prefix sum of one 64-bit value per thread (e.g. four 16-bit counters packed side by side);
// *total = block sum. scratch: >= 32 unsigned long long in LDS.
template <int NT = 0>
__device__ __forceinline__ unsigned long long block_exclusive_scan64(unsigned long long v, unsigned long long* scratch,
                                                                      unsigned long long* total) {
  auto* sl = CFEAR_LDS_PTR(unsigned long long, scratch);
  const int lane = lane_id(), w = threadIdx.x >> 6, nw = NT ? (NT + 63) >> 6 : (blockDim.x + 63) >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  __syncthreads();
  if (lane == 63) sl[w] = inc;
  __syncthreads();
  unsigned long long base = 0, tot = 0;
  for (int i = 0; i < nw; i++) {
    const unsigned long long s = sl[i];
    if (i < w) base += s;
    tot += s;
  }
  *total = tot;
  return base + inc - v;
}

// ---- wave64 sum of a double with DPP moves (no LDS round trips); the total is returned in every lane ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const unsigned long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xF, true);
  return v + __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);  // row_mirror: every lane holds its 16-lane row sum
  v = dpp_add<0x142, 0xA>(v);  // row_bcast15 -> rows 1, 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast31 -> rows 2, 3: lanes 48..63 hold the wave total
  const unsigned long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// ---- wave64 sums of TEN doubles at once, transposed: instead of ten full reductions (six exchange steps each) the lanes
// first split the work - after the exchange with lane ^ 1 a lane carries five of the ten sums, after lane ^ 2 three -
// and only those are carried through the remaining four steps (rotations by 4 and 8 inside the 16-lane rows, then gfx950's
// row and half swaps across the rows): 13 additions per lane instead of 60. On return every lane holds the wave totals of
// its residue class lane % 4; lanes 60..63 are the ones that store them:
//   lane 60: out[0..2] = sums 0, 1, 2   lane 62: out[0..1] = sums 3, 4   lane 61: out[0..2] = sums 5, 6, 7   lane 63: out[0..1] = sums 8, 9
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_get(double v) {
  const unsigned long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xF, true);
  return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ void wave_sum10_transposed(const double (&v)[10], double (&out)[3]) {
  const int lane = lane_id();
  const bool odd = (lane & 1) != 0, hi2 = (lane & 2) != 0;
  double u[5];
#pragma unroll
  for (int k = 0; k < 5; k++) {  // lane ^ 1: even lanes keep sums 0..4, odd lanes 5..9
    const double keep = odd ? v[k + 5] : v[k], give = odd ? v[k] : v[k + 5];
    u[k] = keep + dpp_get<0xB1, 0xF>(give);  // quad_perm [1,0,3,2]
  }
  double w[3];
#pragma unroll
  for (int k = 0; k < 2; k++) {  // lane ^ 2: of a lane's five sums, entries 0, 1 stay where bit 1 is clear, 3, 4 where it is set
    const double keep = hi2 ? u[k + 3] : u[k], give = hi2 ? u[k] : u[k + 3];
    w[k] = keep + dpp_get<0x4E, 0xF>(give);  // quad_perm [2,3,0,1]
  }
  w[2] = u[2] + dpp_get<0x4E, 0xF>(u[2]);    // entry 2 is carried by both
#pragma unroll
  for (int k = 0; k < 3; k++) {
    w[k] += dpp_get<0x124, 0xF>(w[k]);  // row_ror:4: lanes of equal lane % 4 inside a row
    w[k] += dpp_get<0x128, 0xF>(w[k]);  // row_ror:8: every lane holds the row sum of its residue class
    // across the four rows with the lane position kept (a row broadcast would hand every lane the value of lane 15, a
    // different residue class): gfx950's row / half swaps. permlane16_swap(x, x) = {(r0, r0, r2, r2), (r1, r1, r3, r3)}
    {
      const unsigned long long b = __double_as_longlong(w[k]);
      const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
      const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
      w[k] = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
    }
    {  // permlane32_swap(y, y) = {(y.lo, y.lo), (y.hi, y.hi)}
      const unsigned long long b = __double_as_longlong(w[k]);
      const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
      const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
      w[k] = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]) + __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
    }
    out[k] = w[k];  // every lane: the wave total of the sums of its residue class
  }
}

// In-place ascending bitonic sort of keys[0..p2) (p2 = power of two, padded by the caller).
template <typename KeyPtr>
__device__ __forceinline__ void block_bitonic_sort(KeyPtr keys, int p2) {
  for (int k = 2; k <= p2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < (p2 >> 1); t += blockDim.x) {
        // t enumerates the pairs (i, i^j) with i having bit j clear
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int ixj = i | j;
        const uint64_t a = keys[i], b = keys[ixj];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
      }
    }
  }
  __syncthreads();
}

}  // namespace cfear_dev
