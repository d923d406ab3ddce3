// pipeline.hip -- kernels and C-ABI entry points for stages 1.5-3 and the batched odometry step.
// Interface contract + reference citations: include/cfear_hip.h. Device code: features_dev.h,
// registration_dev.h.
//
// One radar sweep of B independent sequences = three launches on the context stream, no host round trip:
//   kstrongest_kernel      (kstrongest.hip)  B*A wavefront-rows, HBM streaming
//   features_step_kernel   one 512-thread workgroup per sequence (79,680 B of LDS, two per compute unit): cloud + motion
//                          compensation, voxel bitmap + counting sort, chunked cell statistics, cell-mean grid
//   register_step_kernel   one 256-thread workgroup per sequence (53 KB LDS incl. the match array, three per compute
//                          unit so that the serial Levenberg-Marquardt controllers of different sequences overlap):
//                          association, robust normal equations, LM, outer loop, keyframe logic
// Per-call entry points (clouds, scans, Register, GetCost, cost-sampling covariance) use the same device code.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <new>

#include "common.h"
#include "odometry_step_dev.h"

namespace {

// ---- per-call kernels -------------------------------------------------------------------------
// one workgroup per cloud: block 0 the filtered cloud, block 1 (if launched) the peaks cloud (radar_driver.cpp:59-60); each also writes its
// cloud's host mirror (cfear_cloud::h_block) when it has one
struct CloudOut { float* xyi; int* d_n; float* h_xyi; int* h_n; int cap; };
__device__ __forceinline__ void cloud_mirror_block(const float* xyi, int n, int cap, float* h_xyi, int* h_n) {
  if (!h_xyi) return;
  __syncthreads();  // the block's own writes of xyi
  const int m = 3 * (n < cap ? n : cap);
  for (int i = threadIdx.x; i < m; i += blockDim.x) h_xyi[i] = xyi[i];
  if (threadIdx.x == 0) *h_n = n;
}
__global__ __launch_bounds__(BLOCK_F) void cloud_kernel(const uint32_t* slots, int A, int k, const double* trig, float rr,
                                                        float md, CloudOut o0, CloudOut o1) {
  __shared__ int red_i[64];
  const int peaks = blockIdx.x;
  const CloudOut o = peaks ? o1 : o0;
  const int n = cloud_build_block(slots, A, k, trig, rr, md, peaks, o.xyi, o.cap, red_i);
  if (threadIdx.x == 0) *o.d_n = n;
  cloud_mirror_block(o.xyi, n, o.cap, o.h_xyi, o.h_n);
}

// Compensate (utils.cpp:96-113) of one cloud or of a sweep's two (the fuser compensates cloud and cloud_peaks by the same motion,
// odometrykeyframefuser.cpp:148-149): one workgroup each
__global__ __launch_bounds__(BLOCK_F) void compensate_kernel(CloudOut o0, CloudOut o1, double m0, double m1, double m2, int ccw) {
  const CloudOut o = blockIdx.x ? o1 : o0;
  const int n = *o.d_n;
  compensate_block(o.xyi, n, m0, m1, m2, ccw);
  cloud_mirror_block(o.xyi, n, o.cap, o.h_xyi, o.h_n);
}

__global__ __launch_bounds__(BLOCK_F) void features_kernel(ScanDev* S, const float* src_xyi, const int* d_n, FeatureParams P,
                                                           BlockScratch B) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[FeatLdsC::total];
  int n = *d_n;
  if (n > S->cap_points) n = S->cap_points;
  int bytes = 1;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float x = src_xyi[3 * i], y = src_xyi[3 * i + 1], w = src_xyi[3 * i + 2];
    S->xyi[3 * i] = x; S->xyi[3 * i + 1] = y; S->xyi[3 * i + 2] = w;
    bytes &= (w >= 0.f && w <= 255.f && w == (float)(int)w) ? 1 : 0;
  }
  const bool byte_intensities = __syncthreads_and(bytes) != 0;  // (the barrier also makes the copy visible to the whole block)
  PointRegs PR;
  point_regs_from_global(S->xyi, n, PR);
  features_dispatch(S, n, P, B, lds, nullptr, nullptr, false, byte_intensities, PR);
}

// MapPointNormal from given cells (raw = true identity cells, pointnormal.cpp:76-82; the transformed-copy constructor
// :91-110): the cells are already in S->cells; this writes their float means, the registration views and the search grid
__global__ __launch_bounds__(BLOCK_F) void scan_from_cells_kernel(ScanDev* S, int n, FeatureParams P, BlockScratch B) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[FeatLdsC::bm];  // the two reduction arrays
  const FeatureScratch W = make_fscratch(B, lds);
  const size_t cc = (size_t)S->cap_cells;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const cfear_cell c = S->cells[i];
    S->mean_f[2 * i] = (float)c.mean[0]; S->mean_f[2 * i + 1] = (float)c.mean[1];
    double* rs = S->rsrc + i;
    rs[0] = c.mean[0]; rs[cc] = c.mean[1]; rs[2 * cc] = c.normal[0]; rs[3 * cc] = c.normal[1]; rs[4 * cc] = (double)c.nsamples; rs[5 * cc] = c.scale;
    double* rt = S->rtar + 8 * (size_t)i;
    rt[0] = c.mean[0]; rt[1] = c.mean[1]; rt[2] = c.normal[0]; rt[3] = c.normal[1]; rt[4] = (double)c.nsamples; rt[5] = c.scale;
    double* rc = S->rcov + 3 * (size_t)i;
    rc[0] = c.cov[0]; rc[1] = c.cov[1]; rc[2] = c.cov[2];
  }
  if (threadIdx.x == 0) { S->n_points = 0; S->n_samples = 0; S->n_cells = n; S->status = n > 0 ? 0 : CFEAR_ERR_EMPTY; }
  __syncthreads();
  cell_grid_block(S, n, P, W, false, nullptr);
  if (P.nn_tie == 2) kd_build_block(S, B);
}

__global__ void closest_kernel(const ScanDev* S, const double* q, int nq, double d, int* idx, int rule) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  if (rule == 0) { idx[i] = scan_closest(grid_view(S), q[2 * i], q[2 * i + 1], d); return; }
  KdVisit stack[CFEAR_KD_STACK];  // (per-thread scratch: this little kernel only)
  idx[i] = scan_closest_rule(S, grid_view(S), q[2 * i], q[2 * i + 1], d, rule, stack);
}

__global__ __launch_bounds__(BLOCK_R, 3) void register_kernel(ScanDev* const* scans, int n, double* poses, double* cov6, RegParams P,
                                                           BlockScratch B, cfear_reg_summary* out, const double* prior_cov6) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[RegLds::total];
  ScanDev** sp = reinterpret_cast<ScanDev**>(lds + RegLds::scanptr);
  for (int i = threadIdx.x; i < n; i += blockDim.x) sp[i] = scans[i];
  __syncthreads();
  const RegScratch W = make_rscratch(B, lds);
  register_block(sp, n, poses, cov6, P, W, reinterpret_cast<double*>(lds + RegLds::par),
                 reinterpret_cast<RegShared*>(lds + RegLds::regsh), out, nullptr, prior_cov6);
}

__global__ __launch_bounds__(BLOCK_R, 3) void get_cost_kernel(ScanDev* const* scans, int n, const double* poses, RegParams P, BlockScratch B,
                                                           int itr, double* score, double* residuals, int cap, int* n_res) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[RegLds::total];
  ScanDev** sp = reinterpret_cast<ScanDev**>(lds + RegLds::scanptr);
  for (int i = threadIdx.x; i < n; i += blockDim.x) sp[i] = scans[i];
  __syncthreads();
  const RegScratch W = make_rscratch(B, lds);
  get_cost_block(sp, n, poses, P, W, reinterpret_cast<double*>(lds + RegLds::par), reinterpret_cast<RegShared*>(lds + RegLds::regsh), itr,
                 score, residuals, cap, n_res);
}

// GetCost of many candidate poses of the last scan at once (cost-sampling covariance, odometrykeyframefuser.cpp:291-321):
// one workgroup per sample pose; match arrays of workgroup b at match_base + b * 8 * cap (used when they do not fit in LDS)
__global__ __launch_bounds__(BLOCK_R, 3) void get_cost_samples_kernel(ScanDev* const* scans, int n, const double* poses, const double* samples,
                                                                   RegParams P, double* match_base, int* assoc_base, int cap, int itr,
                                                                   double* costs, int* n_res) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[RegLds::total];
  // the sample's poses are written where get_cost_block keeps its parameter vectors: thread i converts pose i in place
  double* my_poses = reinterpret_cast<double*>(lds + RegLds::par);
  const int b = blockIdx.x;
  ScanDev** sp = reinterpret_cast<ScanDev**>(lds + RegLds::scanptr);
  for (int i = threadIdx.x; i < n; i += blockDim.x) sp[i] = scans[i];
  for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) my_poses[i] = (i >= 3 * (n - 1)) ? samples[3 * b + (i - 3 * (n - 1))] : poses[i];
  __syncthreads();
  RegScratch W;
  const size_t c = (size_t)cap;
  double* m = match_base + (size_t)b * 8 * c;
  W.tmx = m; W.tmy = m + c; W.a0 = m + 2 * c; W.a1 = m + 3 * c; W.a2 = m + 4 * c; W.sx = m + 5 * c; W.sy = m + 6 * c; W.w = m + 7 * c;
  W.assoc = assoc_base + (size_t)b * 3 * c; W.cap = cap; W.acap = 3 * cap;
  W.red = reinterpret_cast<double*>(lds + RegLds::red_d);
  W.red_i = reinterpret_cast<int*>(lds + RegLds::red_i);
  get_cost_block(sp, n, my_poses, P, W, reinterpret_cast<double*>(lds + RegLds::par), reinterpret_cast<RegShared*>(lds + RegLds::regsh), itr,
                 costs + b, nullptr, 0, n_res + b);
}

// ---- batched odometry: one launch per stage and sweep (bodies: odometry_step_dev.h) ----
// TIMED: per-phase timestamps (tools/); the production instantiation carries no timer at all
template <bool TIMED>
__global__ __launch_bounds__(BLOCK_F, 4) void features_step_kernel(const uint32_t* slots_all, const double* trig, OdoParams OP,
                                                                   const SeqState* states, ScanDev* const* scan_slots,
                                                                   const BlockScratch* scratch) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[FeatLdsC::total];
  features_step_body<TIMED>(lds, OP.seq0 + (int)blockIdx.x, slots_all, trig, OP, states, scratch);
}
// the production registration shape (256 threads, three workgroups per unit) compiled for the 64 scans a sequence can keep: submaps of 8 .. 63
// keyframes when there are more sequences than compute units (launch_register_step). A kernel name of its own: register_step.hip instantiates
// register_step_kernel with 8-scan shared state under the same template arguments, and profilers key on the name.
template <bool TIMED, int KCOST>
__global__ __launch_bounds__(BLOCK_R, CFEAR_REG_MIN_WG) void register_step64_kernel(OdoParams OP, SeqState* states, const BlockScratch* scratch, double* cov_work,
                                                                                   cfear_reg_summary* summaries, double* poses_out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[RegLds::total];
  register_step_body<TIMED, KCOST>(lds, OP.order ? OP.order[blockIdx.x] : OP.seq0 + (int)blockIdx.x, OP, states, scratch, cov_work, summaries, poses_out);
}
// the same stage from clouds on the device (filter_type CA-CFAR / cfear_odometry_step_cloud_device)
__global__ __launch_bounds__(BLOCK_F, 4) void features_cloud_step_kernel(const float* xyi_all, int cap, const int* counts, OdoParams OP,
                                                                         const SeqState* states, const BlockScratch* scratch) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[FeatLdsC::total];
  features_cloud_step_body(lds, OP.seq0 + (int)blockIdx.x, xyi_all, cap, counts, OP, states, scratch);
}
// Registration workgroups longest first: sequences ordered by the work their registration took in the previous sweep (a sequence's
// scene changes slowly), as a counting sort over 256 buckets of the key - one workgroup, a few microseconds. The order inside a
// bucket is whatever the atomics give: results do not depend on which workgroup slot a sequence takes.
__global__ __launch_bounds__(1024) void order_kernel(const unsigned* work, int B, int* order) {
  __shared__ unsigned hist[256], red[16];
  const int tid = threadIdx.x;
  unsigned mx = 1u;
  for (int i = tid; i < B; i += 1024) mx = max(mx, work[i]);
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  if (tid < 256) hist[tid] = 0u;
  __syncthreads();
  mx = 1u;
  for (int i = 0; i < 16; i++) mx = max(mx, red[i]);
  const float scale = 255.0f / (float)mx;
  for (int i = tid; i < B; i += 1024) atomicAdd(&hist[255 - min(255, (int)((float)work[i] * scale))], 1u);  // bucket 0 = the most work
  __syncthreads();
  if (tid < 64) {  // exclusive scan of the 256 counts: four per lane, a wave scan across the lanes
    const unsigned c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
    unsigned v = c0 + c1 + c2 + c3, incl = v;
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)incl, o); if (tid >= o) incl += u; }
    const unsigned ex = incl - v;
    hist[4 * tid] = ex; hist[4 * tid + 1] = ex + c0; hist[4 * tid + 2] = ex + c0 + c1; hist[4 * tid + 3] = ex + c0 + c1 + c2;
  }
  __syncthreads();
  for (int i = tid; i < B; i += 1024) order[atomicAdd(&hist[255 - min(255, (int)((float)work[i] * scale))], 1u)] = i;
}
// ---- host-side helpers ---------------------------------------------------------------------------
size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

FeatureParams feature_params(const cfear_ctx* ctx) {
  FeatureParams P;
  P.nn_tie = ctx->tune_nn_tie;
  P.range_res = ctx->par.range_res; P.min_distance = ctx->par.min_distance;
  P.radius = (float)ctx->par.res;
  P.downsample_factor = ctx->par.downsample_factor;
  P.weight_intensity = ctx->par.weight_intensity;
  P.assoc_radius = ctx->par.assoc_radius;
  return P;
}
RegParams reg_params(const cfear_ctx* ctx) {
  RegParams P;
  P.cost = ctx->par.cost; P.loss = ctx->par.loss; P.weight_opt = ctx->par.weight_opt;
  P.recompute_repeats = ctx->tune_repeat_shortcut ? 0 : 1;
  P.nn_tie = ctx->tune_nn_tie;
  P.loss_limit = ctx->par.loss_limit; P.covar_scale = ctx->par.covar_scale; P.regularization = ctx->par.regularization;
  P.assoc_radius = ctx->par.assoc_radius;
  P.max_outer = ctx->par.max_itr_association; P.min_itr = ctx->par.min_itr; P.max_inner = ctx->par.max_solver_iterations;
  return P;
}

constexpr int GRID_CAP = CFEAR_GRID_CAP;

struct ScanLayout { size_t xyi, cells, mean_f, gstart, gpts, rsrc, rtar, rcov, kd_nodes, kd_vind, kd_data, total; };
// cap_cells <= cap_points: cells a scan can hold (every cell is the centroid neighbourhood of an occupied voxel, so never more than
// points; the batched odometry may be sized for fewer: cfear_tune MAX_CELLS). with_cells: the 120-byte cfear_cell records exist (the
// per-call scans, whose cells can be downloaded); the scans of the batched odometry objects go without
ScanLayout scan_layout(int cap_points, int cap_cells, bool with_cells = true, bool with_kd = false) {
  ScanLayout L;
  size_t o = align_up(sizeof(ScanDev), 256);
  L.xyi = o; o = align_up(o + sizeof(float) * 3 * (size_t)cap_points, 256);
  L.cells = o; if (with_cells) o = align_up(o + sizeof(cfear_cell) * (size_t)cap_cells, 256);
  L.mean_f = o; o = align_up(o + sizeof(float) * 2 * (size_t)cap_cells, 256);
  L.gstart = o; o = align_up(o + (sizeof(int) + sizeof(uint2)) * (GRID_CAP + 4), 256);  // 32-bit offsets + the region of the 16-bit ones (grid_off16)
  L.gpts = o; o = align_up(o + sizeof(float4) * (size_t)cap_cells, 256);
  L.rsrc = o; o = align_up(o + sizeof(double) * 6 * (size_t)cap_cells, 256);
  L.rtar = o; o = align_up(o + sizeof(double) * 8 * (size_t)cap_cells, 256);
  L.rcov = o; o = align_up(o + sizeof(double) * 3 * (size_t)cap_cells, 256);
  L.kd_nodes = L.kd_vind = L.kd_data = 0;
  if (with_kd) {  // cfear_tune NN_TIE_RULE = 2 (kdtree_flann_dev.h)
    L.kd_nodes = o; o = align_up(o + sizeof(KdNode) * (2 * (size_t)cap_cells + 2), 256);
    L.kd_vind = o; o = align_up(o + sizeof(int) * (size_t)cap_cells, 256);
    L.kd_data = o; o = align_up(o + sizeof(float) * 2 * (size_t)cap_cells, 256);
  }
  L.total = o;
  return L;
}
// writes a ScanDev header for a flat device block at d_base
ScanDev scan_header(unsigned char* d_base, int cap_points, int cap_cells, bool with_cells = true, bool with_kd = false) {
  const ScanLayout L = scan_layout(cap_points, cap_cells, with_cells, with_kd);
  ScanDev h;
  memset(&h, 0, sizeof(h));
  h.status = CFEAR_ERR_EMPTY;
  h.cap_points = cap_points; h.cap_cells = cap_cells; h.cap_grid = GRID_CAP;
  h.xyi = reinterpret_cast<float*>(d_base + L.xyi);
  h.cells = with_cells ? reinterpret_cast<cfear_cell*>(d_base + L.cells) : nullptr;
  h.rcov = reinterpret_cast<double*>(d_base + L.rcov);
  if (with_kd) {
    h.kd.nodes = reinterpret_cast<KdNode*>(d_base + L.kd_nodes); h.kd.vind = reinterpret_cast<int*>(d_base + L.kd_vind);
    h.kd.data = reinterpret_cast<float*>(d_base + L.kd_data);
  }
  h.kd.root = -1;
  h.mean_f = reinterpret_cast<float*>(d_base + L.mean_f);
  h.gstart = reinterpret_cast<int*>(d_base + L.gstart);
  h.gpts = reinterpret_cast<float4*>(d_base + L.gpts);
  h.rsrc = reinterpret_cast<double*>(d_base + L.rsrc);
  h.rtar = reinterpret_cast<double*>(d_base + L.rtar);
  h.gcell = 1.f;
  return h;
}

struct ScratchLayout { size_t keys, spts, order, vstart, vlist, vcur, rng, part, tmpi, samples, match, assoc, total; int p2cap; bool big; };
ScratchLayout scratch_layout(int cap_points, int pair_cap) {
  ScratchLayout L;
  int p2 = 1; while (p2 < cap_points) p2 <<= 1;
  L.p2cap = p2;
  L.big = true;  // the general feature path (clouds beyond the compact path's limits) works in these global arrays
  const size_t cp = (size_t)cap_points, kp = (size_t)p2;
  size_t o = 0;
  L.keys = o; o = align_up(o + sizeof(uint64_t) * kp, 256);
  L.spts = o; o = align_up(o + sizeof(float) * 3 * cp, 256);
  L.order = o; o = align_up(o + sizeof(int) * cp, 256);
  L.vstart = o; o = align_up(o + sizeof(int) * (cp + 2), 256);
  L.vlist = o; o = align_up(o + sizeof(int) * cp, 256);
  L.vcur = o; o = align_up(o + sizeof(int) * (GRID_CAP + 4), 256);
  L.rng = o; o = align_up(o + sizeof(int) * 8 * (size_t)cap_points, 256);
  L.part = o; o = align_up(o + sizeof(double) * 7 * (size_t)cap_points, 256);
  L.tmpi = o; o = align_up(o + sizeof(int) * (2 * (size_t)cap_points + 16), 256);
  L.samples = o; o = align_up(o + sizeof(float) * 3 * (size_t)cap_points, 256);
  L.match = o; o = align_up(o + sizeof(double) * 8 * (size_t)pair_cap, 256);
  L.assoc = o; o = align_up(o + sizeof(int) * 3 * (size_t)pair_cap, 256);  // registration_dev.h build_problem_block: six ints per (group of four keyframes, source cell) <= 3 per pair
  L.total = o;
  return L;
}
BlockScratch scratch_header(unsigned char* d_base, int cap_points, int pair_cap) {
  const ScratchLayout L = scratch_layout(cap_points, pair_cap);
  BlockScratch B;
  B.keys = reinterpret_cast<uint64_t*>(d_base + L.keys);
  B.spts = reinterpret_cast<float*>(d_base + L.spts);
  B.order = reinterpret_cast<int*>(d_base + L.order);
  B.vstart = reinterpret_cast<int*>(d_base + L.vstart);
  B.vlist = reinterpret_cast<int*>(d_base + L.vlist);
  B.vcur = reinterpret_cast<int*>(d_base + L.vcur);
  B.rng = reinterpret_cast<int*>(d_base + L.rng);
  B.part = reinterpret_cast<double*>(d_base + L.part);
  B.tmpi = reinterpret_cast<int*>(d_base + L.tmpi);
  B.samples = reinterpret_cast<float*>(d_base + L.samples);
  B.match = reinterpret_cast<double*>(d_base + L.match);
  B.assoc = reinterpret_cast<int*>(d_base + L.assoc);
  B.cap_points = cap_points; B.p2cap = L.p2cap; B.pair_cap = pair_cap;
  B.vrank = nullptr; B.vperm = nullptr;
  return B;
}

int set_kernel_attributes(cfear_ctx*) { return CFEAR_OK; }  // LDS is static (up to 160 KiB per workgroup on gfx950)

}  // namespace

struct cfear_scan {
  unsigned char* d_block = nullptr;  // ScanDev header + arrays (a block of the context's pool)
  size_t bytes = 0;
  int n_cells = -1;                  // known on the host since creation (cfear_scan_size without a device round trip)
  int cap_points = 0;
  bool with_kd = false;  // built under cfear_tune NN_TIE_RULE = 2: carries the kd-tree the parity mode's search walks
};
// cfear_tune NN_TIE_RULE = 2 needs scans that were built in that mode: anything else would answer by the production rule without saying so
static int check_tie_rule_scans(cfear_ctx* ctx, cfear_scan* const* scans, int n, const char* what) {
  if (ctx->tune_nn_tie != 2) return CFEAR_OK;
  for (int i = 0; i < n; i++)
    if (scans[i] && !scans[i]->with_kd) {
      char msg[256];
      snprintf(msg, sizeof(msg), "%s: scan %d was created before cfear_tune NN_TIE_RULE = 2 was set - it has no kd-tree for the parity mode's search; create the scans after the tune call", what, i);
      return cfear_fail(ctx, CFEAR_ERR_INVALID, msg);
    }
  return CFEAR_OK;
}
struct cfear_odometry {
  int B = 0, nslots = 0, cap_points = 0, cap_cells = 0, pair_cap = 0;
  int* d_order = nullptr; unsigned* d_work = nullptr;  // registration workgroups longest first (cfear_tune REGISTRATION_ORDER): see order_kernel
  bool order_ready = false;  // d_work holds the keys of a registration launch
  bool with_kd = false;  // the scans carry FLANN kd-tree arrays (cfear_tune NN_TIE_RULE = 2 at creation)
  int large_kernel = 0, n_cus = 256;  // cfear_tune LARGE_SUBMAP_KERNEL at creation; compute units of the device
  int* d_flags = nullptr;  // bit 0: a scan had more cells than cap_cells, bit 1: a cloud had more points than cap_points (only allocated when either can happen)
  unsigned char* d_scans = nullptr;    // B * nslots flat scan blocks
  ScanDev** d_scan_ptrs = nullptr;     // [B * nslots]
  size_t scan_stride = 0;              // bytes between consecutive scan slots of d_scans
  unsigned char* d_scratch = nullptr;  // B scratch blocks
  BlockScratch* d_scratch_hdr = nullptr;
  SeqState* d_states = nullptr;
  double* d_poses_work = nullptr;
  double* d_cov_work = nullptr;
  cfear_reg_summary* d_summaries = nullptr;
  double* d_poses_out = nullptr;
  uint32_t* d_slots[2] = {nullptr, nullptr};  // filter output, double-buffered: the filter runs one sweep ahead
  uint8_t* d_polar = nullptr;  // staging for step_host
  // filter_type CA-CFAR (radar_driver.cpp:52-56): the filter's output is a cloud per sequence instead of A * k slots
  int filter = CFEAR_FILTER_KSTRONG;
  float* d_cloud = nullptr;      // [B][cap_points][3]
  int* d_cloud_n = nullptr;      // [B] detections per sequence (may exceed cap_points: the cloud keeps the first cap_points)
  int* d_cfar_rows = nullptr;    // row counts / row bases / hit masks of one sweep (cfear_cfar_scratch_ints)
  float* rp_cloud[2] = {nullptr, nullptr};  // replay: clouds of a chunk of sweeps, double-buffered like rp_slots
  int* rp_cloud_n[2] = {nullptr, nullptr};
  int* rp_cfar_rows = nullptr;   // ... of a chunk (the replay stream runs one filter at a time)
  // cfear_odometry_replay_host: chunks of sweeps are copied and filtered on a stream of their own (rp_stream), two chunks
  // in flight (staging + slots double-buffered), while the context stream runs features -> registration sweep after sweep
  hipStream_t rp_stream = nullptr;
  uint8_t* rp_polar[2] = {nullptr, nullptr};
  uint32_t* rp_slots[2] = {nullptr, nullptr};
  hipEvent_t rp_filt[2] = {nullptr, nullptr}, rp_used[2] = {nullptr, nullptr};  // chunk filtered / chunk consumed by the odometry kernels
  hipEvent_t rp_in = nullptr;  // device-resident frames ready on the context stream
  bool rp_used_pending[2] = {false, false};
  bool rp_ready = false;       // stream + events exist
  int rp_chunk = 0;            // sweeps per chunk the slot buffers are sized for
  int rp_polar_chunk = 0;      // ... and the staging buffers (allocated on the first replay from host memory only)
  cfear_sweep_record* d_records = nullptr;
  size_t records_cap = 0;      // records
  long long* d_phase_times = nullptr;  // optional [B][32] (cfear_odometry_phase_times)
  int phase_detail = 1;
  bool wg_only = false;  // the table only receives the workgroups' start / end clocks, from the production kernels
  // The filter of a sweep needs nothing but its input, and it is bound by HBM while features / registration are chains
  // of short dependent phases. With overlap on it runs on a stream of its own (sf) into the slot buffer of its parity;
  // features / registration follow on a second stream (so) once their slot buffer is written (ev_filt) and release it
  // when features has consumed it (ev_free). The host issues step t+1 while so still works on step t, so filter(t+1)
  // can start on compute units that the last rounds of registration(t) leave idle. The context stream joins in the
  // reading calls (odo_join). With overlap off (the default, see DESIGN.md: the three kernels want the same registers and
  // LDS of a compute unit, so running them side by side stretches all of them) they run in turn on the context stream.
  int overlap = 0;  // 0: the three kernels in turn on the context stream; n >= 1: filter stream + n odometry streams (below)
  long long step_no = 0;
  // overlap = n: the filter runs on a LOW-priority stream of its own (sf) into the slot buffer of its parity, one sweep ahead;
  // the sequences are cut into n contiguous ranges whose features / registration kernels run on n HIGH-priority streams
  // (so[i]). Different ranges are at different points of their features -> registration chain, so the ragged last round of one
  // kernel is filled with workgroups of another (a features workgroup and a registration workgroup fit a compute unit
  // together), and the filter takes what is left.
  hipStream_t sf = nullptr;
  std::vector<hipStream_t> so;
  hipEvent_t ev_in = nullptr, ev_copied = nullptr;
  hipEvent_t ev_filt[2] = {nullptr, nullptr};
  std::vector<hipEvent_t> ev_free[2], ev_done;  // per odometry stream
  bool filt_pending[2] = {false, false};  // ev_filt / ev_free have been recorded at least once
  // profiling: timing events taken from a pool that is created up front (and grown in blocks)
  bool profile = false;
  std::vector<hipEvent_t> pool;
  size_t pool_used = 0;
  std::vector<hipEvent_t> filter_events;  // 2 per profiled filter launch (borrowed from the pool)
  std::vector<hipEvent_t> stage_events;   // 3 per profiled step: before features, between, after registration
};
// a timing event from the pool, recorded on `st`
static int odo_timed_event(cfear_ctx* ctx, cfear_odometry* o, std::vector<hipEvent_t>& list, hipStream_t st) {
  if (o->pool_used == o->pool.size()) {
    for (int i = 0; i < 1024; i++) {
      hipEvent_t e = nullptr;
      CFEAR_HIP_CHECK(ctx, hipEventCreate(&e));
      o->pool.push_back(e);
    }
  }
  hipEvent_t e = o->pool[o->pool_used++];
  list.push_back(e);
  CFEAR_HIP_CHECK(ctx, hipEventRecord(e, st));
  return CFEAR_OK;
}
static int odo_capacity_check(cfear_ctx* ctx, cfear_odometry* o, const char* what);
// make everything the internal streams have been given so far visible to the context stream
static int odo_join(cfear_ctx* ctx, cfear_odometry* o) {
  if (o->overlap && o->step_no > 0) {
    for (size_t i = 0; i < o->so.size(); i++) {  // an odometry stream waits for every filter it consumes: joining them joins sf
      CFEAR_HIP_CHECK(ctx, hipEventRecord(o->ev_done[i], o->so[i]));
      CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, o->ev_done[i], 0));
    }
  }
  return CFEAR_OK;
}

// replay.hip: features -> registration of `cnt` consecutive sweeps of every sequence in one launch (a persistent workgroup per
// sequence); odo_params points at an OdoParams (the struct is local to each translation unit, same definition)
// register_step.hip: the batched registration step kernel for registrations of up to CFEAR_STEP_SMALL_SCANS scans
__attribute__((visibility("hidden"))) void cfear_launch_register_step_small(const void* odo_params, int count, hipStream_t st, void* states, void* const* scan_slots,
                                                                           const void* scratch, double* poses_work, double* cov_work,
                                                                           cfear_reg_summary* summaries, double* poses_out);
// register_step_large.hip: ... of more scans (submap_scan_size 8 .. 63)
__attribute__((visibility("hidden"))) void cfear_launch_register_step_large(const void* odo_params, int count, hipStream_t st, void* states, const void* scratch,
                                                                           double* cov_work, cfear_reg_summary* summaries, double* poses_out);
__attribute__((visibility("hidden"))) void cfear_launch_replay_chunk(const uint32_t* d_slots, int cnt, int B, const double* d_trig, const void* odo_params,
                                                                    void* states, const void* scratch, double* cov_work, cfear_reg_summary* summaries,
                                                                    double* poses_out, cfear_sweep_record* records, hipStream_t stream);
// ... from the clouds of a chunk ([cnt][B][cap][3] floats, [cnt][B] counts) instead of slots
__attribute__((visibility("hidden"))) void cfear_launch_replay_chunk_cloud(const float* d_xyi, int cap, const int* d_counts, int cnt, int B, const void* odo_params,
                                                                          void* states, const void* scratch, double* cov_work, cfear_reg_summary* summaries,
                                                                          double* poses_out, cfear_sweep_record* records, hipStream_t stream);
// has the object the shape the context's parameters ask for? (k_strongest / submap_scan_size / the filter cannot change under an object)
static bool odo_shape_ok(const cfear_ctx* ctx, const cfear_odometry* o);

// the kernel parameters of one odometry step of `o` under the context's current settings
static OdoParams odo_params(const cfear_ctx* ctx, const cfear_odometry* o) {
  OdoParams OP;
  OP.fp = feature_params(ctx); OP.rp = reg_params(ctx);
  OP.A = ctx->A; OP.k = ctx->par.k_strongest; OP.compensate = ctx->par.compensate; OP.ccw = ctx->par.radar_ccw;
  OP.use_keyframe = ctx->par.use_keyframe; OP.submap = ctx->par.submap_scan_size;
  OP.min_keyframe_dist = ctx->par.min_keyframe_dist; OP.min_keyframe_rot_deg = ctx->par.min_keyframe_rot_deg;
  OP.phase_times = o->wg_only ? nullptr : o->d_phase_times; OP.phase_detail = o->phase_detail;
  OP.wg_times = o->wg_only ? o->d_phase_times : nullptr;
  OP.seq0 = 0;
  OP.scans_base = o->d_scans; OP.scan_stride = o->scan_stride;
  OP.records = nullptr;
  OP.flags = o->d_flags;
  OP.order = nullptr; OP.work = o->d_work;
  return OP;
}
// features -> registration of one sweep of every sequence on `st`, from the filter's slots
// the registration step kernel of a sweep. register_step.hip holds the production instantiations (one per cost metric, registrations of
// up to CFEAR_STEP_SMALL_SCANS scans: a bigger LDS match array); a larger submap runs the instantiation of this file (any cost, 64 scans)
static void launch_register_step(const OdoParams& P_in, int count, hipStream_t st, cfear_odometry* o) {
  OdoParams P = P_in;
  if (o->d_order && count == o->B && P.seq0 == 0) {  // (whole-batch launches only: the sub-batches of the overlap mode keep their ranges)
    if (o->order_ready) {
      hipLaunchKernelGGL(order_kernel, dim3(1), dim3(1024), 0, st, o->d_work, o->B, o->d_order);
      P.order = o->d_order;
    }
    o->order_ready = true;  // this launch records the keys of the next one
  }
  if (P.submap + 1 <= CFEAR_STEP_SMALL_SCANS) {
    cfear_launch_register_step_small(&P, count, st, o->d_states, reinterpret_cast<void* const*>(o->d_scan_ptrs), o->d_scratch_hdr, o->d_poses_work, o->d_cov_work, o->d_summaries, o->d_poses_out);
    return;
  }
  // a larger submap (8 .. 63 keyframes). Few sequences - at most one per compute unit - or a very large submap (>= 24 keyframes):
  // register_step_large.hip, a unit's threads and LDS for each registration (2 x faster per registration at fifty keyframes); otherwise the
  // production shape compiled for 64 scans, three workgroups per unit. In aggregate both shapes are bound by the same thing, the vector
  // instructions of the evaluation and the association (0.7-0.8 of the issue slots at fifty keyframes), so they differ by
  // < 10 % at 768 sequences and the small shape is the better one at ten keyframes. cfear_tune LARGE_SUBMAP_KERNEL forces either.
  const bool large = o->large_kernel == 2 || (o->large_kernel == 0 && (count <= o->n_cus || P.submap >= 24));
  if (large) {
    cfear_launch_register_step_large(&P, count, st, o->d_states, o->d_scratch_hdr, o->d_cov_work, o->d_summaries, o->d_poses_out);
    return;
  }
#define CFEAR_LAUNCH_REG(T, C) hipLaunchKernelGGL((register_step64_kernel<T, C>), dim3(count), dim3(BLOCK_R), 0, st, P, o->d_states, o->d_scratch_hdr, \
                                                  o->d_cov_work, o->d_summaries, o->d_poses_out)
  // one instantiation per cost metric here too (the evaluation inline, no run-time dispatch): a ten- or fifty-keyframe submap
  // evaluates thousands of residual blocks 30-80 times per registration
  if (P.phase_times) CFEAR_LAUNCH_REG(true, -1);
  else if (P.rp.cost == CFEAR_COST_P2L) CFEAR_LAUNCH_REG(false, CFEAR_COST_P2L);
  else if (P.rp.cost == CFEAR_COST_P2D) CFEAR_LAUNCH_REG(false, CFEAR_COST_P2D);
  else CFEAR_LAUNCH_REG(false, CFEAR_COST_P2P);
#undef CFEAR_LAUNCH_REG
}
static void odo_launch_sweep(const cfear_ctx* ctx, cfear_odometry* o, const OdoParams& P, const uint32_t* d_slots, int seq_count, hipStream_t st) {
  if (P.phase_times)
    hipLaunchKernelGGL(features_step_kernel<true>, dim3(seq_count), dim3(BLOCK_F), 0, st, d_slots, ctx->d_trig, P, o->d_states,
                       o->d_scan_ptrs, o->d_scratch_hdr);
  else
    hipLaunchKernelGGL(features_step_kernel<false>, dim3(seq_count), dim3(BLOCK_F), 0, st, d_slots, ctx->d_trig, P, o->d_states,
                       o->d_scan_ptrs, o->d_scratch_hdr);
  launch_register_step(P, seq_count, st, o);
}

static void odo_launch_sweep_cloud(cfear_odometry* o, const OdoParams& P, const float* d_xyi, int cap, const int* d_counts, int seq_count, hipStream_t st) {
  hipLaunchKernelGGL(features_cloud_step_kernel, dim3(seq_count), dim3(BLOCK_F), 0, st, d_xyi, cap, d_counts, P, o->d_states, o->d_scratch_hdr);
  launch_register_step(P, seq_count, st, o);
}
static int odo_cfar_points(const cfear_ctx* ctx) { return ctx->par.cfar_max_points > 0 ? ctx->par.cfar_max_points : 32768; }
static bool odo_shape_ok(const cfear_ctx* ctx, const cfear_odometry* o) {
  if (o->nslots != ctx->par.submap_scan_size + 1 || o->filter != ctx->par.filter_type) return false;
  if ((ctx->tune_nn_tie == 2 && !o->with_kd) || (ctx->tune_nn_tie != 0 && o->pair_cap < 8192)) return false;  // the tie rule was switched after the object was created
  if (ctx->tune_voxel_order != 0) return false;  // per-call scans only (cfear_odometry_create refuses it too)
  return o->cap_points == (o->filter == CFEAR_FILTER_CACFAR ? odo_cfar_points(ctx) : ctx->A * ctx->par.k_strongest);
}
// the CA-CFAR stage of n_scans sweeps (radar_driver.cpp:52-56) into clouds of o->cap_points points each
static int odo_launch_cfar(cfear_ctx* ctx, cfear_odometry* o, const uint8_t* d_polar, int n_scans, float* d_xyi, int* d_counts, int* d_rows, hipStream_t st) {
  return cfear_launch_cfar_batch(ctx, d_polar, n_scans, ctx->par.cfar_window_size, ctx->par.cfar_nb_guard_cells, ctx->par.cfar_false_alarm_rate,
                                 ctx->par.cfar_max_distance, d_xyi, o->cap_points, d_counts, d_rows, st);
}

// per-context scratch of the per-call API, sized for up to MAX_SCANS-1 keyframes
static int ensure_ctx_scratch(cfear_ctx* ctx, int cap_points, int pair_cap) {
  const ScratchLayout L = scratch_layout(cap_points, pair_cap);
  const size_t extra = 4096;  // summary + poses + cov + scan pointer table + closest buffers live after the layout
  const size_t need = L.total + extra + sizeof(double) * (3 * MAX_SCANS + 36) + sizeof(void*) * MAX_SCANS + sizeof(cfear_reg_summary);
  if (need > ctx->scratch_bytes) {
    if (ctx->d_scratch) (void)hipFree(ctx->d_scratch);
    ctx->d_scratch = nullptr; ctx->scratch_bytes = 0;
    if (hipMalloc(&ctx->d_scratch, need) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc context scratch");
    if (hipMemset(ctx->d_scratch, 0, need) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_HIP, "hipMemset context scratch");  // dense voxel table starts all-zero
    ctx->scratch_bytes = need;
  }
  return CFEAR_OK;
}

extern "C" {

// ---- clouds ------------------------------------------------------------------------------------
__attribute__((visibility("hidden"))) int cfear_cloud_alloc(cfear_ctx* ctx, int cap, cfear_cloud** out) {
  cfear_cloud* c = new (std::nothrow) cfear_cloud();
  if (!c) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "cloud alloc");
  c->cap = cap > 0 ? cap : 1;
  const int rc = cfear_pool_alloc(ctx, 16 + sizeof(float) * 3 * (size_t)c->cap, &c->block, &c->bytes);  // one block from the context's pool
  if (rc != CFEAR_OK) { delete c; return rc; }
  c->d_n = static_cast<int*>(c->block);
  c->d_xyi = reinterpret_cast<float*>(static_cast<unsigned char*>(c->block) + 16);
  c->h_block = static_cast<unsigned char*>(cfear_hpool_alloc(ctx, 16 + sizeof(float) * 3 * (size_t)c->cap, &c->h_bytes));
  *out = c;
  return CFEAR_OK;
}

void cfear_cloud_release(cfear_ctx* ctx, cfear_cloud* c) {
  if (!c) return;
  cfear_pool_free(ctx, c->block, c->bytes);  // (stream-ordered reuse; no synchronisation, no hipFree on the per-sweep path)
  cfear_hpool_free(ctx, c->h_block, c->h_bytes);
  delete c;
}
static CloudOut cloud_out(cfear_cloud* c) {
  CloudOut o; o.xyi = c->d_xyi; o.d_n = c->d_n; o.cap = c->cap;
  o.h_xyi = c->h_block ? c->h_xyi() : nullptr; o.h_n = c->h_block ? c->h_n() : nullptr;
  return o;
}

int cfear_filter_polar_device(cfear_ctx* ctx, const uint8_t* d_polar, cfear_cloud** cloud, cfear_cloud** cloud_peaks) {
  if (!ctx || !d_polar || !cloud) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_polar: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  *cloud = nullptr;
  if (cloud_peaks) *cloud_peaks = nullptr;
  int rc = cfear_ensure_staging(ctx, 1);
  if (rc != CFEAR_OK) return rc;
  rc = cfear_launch_kstrongest(ctx, d_polar, 1, ctx->d_slots, ctx->stream);  // radar_driver.cpp:58
  if (rc != CFEAR_OK) return rc;
  const int A = ctx->A, k = ctx->par.k_strongest, cap = A * k;
  cfear_cloud *c0 = nullptr, *c1 = nullptr;  // radar_driver.cpp:59-60
  rc = cfear_cloud_alloc(ctx, cap, &c0);
  if (rc != CFEAR_OK) return rc;
  if (cloud_peaks) { rc = cfear_cloud_alloc(ctx, cap, &c1); if (rc != CFEAR_OK) { cfear_cloud_release(ctx, c0); return rc; } }
  hipLaunchKernelGGL(cloud_kernel, dim3(c1 ? 2 : 1), dim3(BLOCK_F), 0, ctx->stream, ctx->d_slots, A, k, ctx->d_trig,
                     ctx->par.range_res, ctx->par.min_distance, cloud_out(c0), cloud_out(c1 ? c1 : c0));
  c0->mirror_valid = c0->h_block != nullptr;
  *cloud = c0;
  if (c1) { c1->mirror_valid = c1->h_block != nullptr; *cloud_peaks = c1; }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

int cfear_filter_polar(cfear_ctx* ctx, const uint8_t* h_polar, cfear_cloud** cloud, cfear_cloud** cloud_peaks) {
  if (!ctx || !h_polar || !cloud) return cfear_fail(ctx, CFEAR_ERR_INVALID, "filter_polar: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = cfear_ensure_staging(ctx, 1);
  if (rc != CFEAR_OK) return rc;
  rc = cfear_upload_image(ctx, ctx->d_polar, h_polar, (size_t)ctx->A * ctx->R);
  if (rc != CFEAR_OK) return rc;
  return cfear_filter_polar_device(ctx, ctx->d_polar, cloud, cloud_peaks);
}

int cfear_cloud_upload(cfear_ctx* ctx, const float* xyi, int n, cfear_cloud** cloud) {
  if (!ctx || !cloud || n < 0 || (n > 0 && !xyi)) return cfear_fail(ctx, CFEAR_ERR_INVALID, "cloud_upload: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  cfear_cloud* c = nullptr;
  int rc = cfear_cloud_alloc(ctx, n, &c);
  if (rc != CFEAR_OK) return rc;
  hipError_t e = hipSuccess;
  if (n > 0) e = hipMemcpyAsync(c->d_xyi, xyi, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(c->d_n, &n, sizeof(int), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { cfear_cloud_release(ctx, c); return cfear_fail(ctx, CFEAR_ERR_HIP, "cloud_upload", e); }
  if (c->h_block) { *c->h_n() = n; if (n > 0) memcpy(c->h_xyi(), xyi, sizeof(float) * 3 * (size_t)n); c->mirror_valid = true; }
  *cloud = c;
  return CFEAR_OK;
}

int cfear_cloud_size(cfear_ctx* ctx, const cfear_cloud* c, int* n) {
  if (!ctx || !c || !n) return cfear_fail(ctx, CFEAR_ERR_INVALID, "cloud_size: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (c->mirror_valid) { CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream)); *n = *c->h_n(); return CFEAR_OK; }
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(n, c->d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

// m clouds to the host with ONE synchronisation: count and points of every cloud travel together into pinned staging, the first
// min(n, capacity) points are handed on (the per-sweep route downloads cloud and cloud_peaks, radar_driver.cpp:59-60 / utils.cpp:96-113)
int cfear_clouds_download(cfear_ctx* ctx, const cfear_cloud* const* clouds, int m, float* const* xyi, const int* capacity, int* n) {
  if (!ctx || !clouds || m <= 0 || m > 16) return cfear_fail(ctx, CFEAR_ERR_INVALID, "clouds_download: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  size_t off[17]; off[0] = 0;
  for (int i = 0; i < m; i++) {
    if (!clouds[i]) return cfear_fail(ctx, CFEAR_ERR_INVALID, "clouds_download: null cloud");
    const int want = (xyi && xyi[i] && capacity) ? (capacity[i] < clouds[i]->cap ? capacity[i] : clouds[i]->cap) : 0;
    off[i + 1] = off[i] + ((16 + sizeof(float) * 3 * (size_t)(want > 0 ? want : 0) + 63) & ~(size_t)63);
  }
  bool all_mirrored = true;
  for (int i = 0; i < m; i++) all_mirrored = all_mirrored && clouds[i]->mirror_valid;
  if (all_mirrored) {  // the kernels wrote the host copies: wait for them, hand the points on
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < m; i++) {
      const int cnt = *clouds[i]->h_n();
      if (n) n[i] = cnt;
      const int want = (xyi && xyi[i] && capacity) ? (capacity[i] < clouds[i]->cap ? capacity[i] : clouds[i]->cap) : 0;
      const int take = cnt < want ? cnt : want;
      if (take > 0) memcpy(xyi[i], clouds[i]->h_xyi(), sizeof(float) * 3 * (size_t)take);
    }
    return CFEAR_OK;
  }
  int rc = cfear_ensure_hstage(ctx, off[m]);
  if (rc != CFEAR_OK) return rc;
  for (int i = 0; i < m; i++) {
    const int want = (xyi && xyi[i] && capacity) ? (capacity[i] < clouds[i]->cap ? capacity[i] : clouds[i]->cap) : 0;
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_stage + off[i], clouds[i]->block, 16 + sizeof(float) * 3 * (size_t)(want > 0 ? want : 0), hipMemcpyDeviceToHost, ctx->stream));
  }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < m; i++) {
    const int cnt = *reinterpret_cast<const int*>(ctx->h_stage + off[i]);
    if (n) n[i] = cnt;
    const int want = (xyi && xyi[i] && capacity) ? (capacity[i] < clouds[i]->cap ? capacity[i] : clouds[i]->cap) : 0;
    const int take = cnt < want ? cnt : want;
    if (take > 0) memcpy(xyi[i], ctx->h_stage + off[i] + 16, sizeof(float) * 3 * (size_t)take);
  }
  return CFEAR_OK;
}

int cfear_cloud_download(cfear_ctx* ctx, const cfear_cloud* c, float* xyi, int capacity, int* n) {
  if (!ctx || !c) return cfear_fail(ctx, CFEAR_ERR_INVALID, "cloud_download: bad argument");
  return cfear_clouds_download(ctx, &c, 1, &xyi, &capacity, n);
}

int cfear_compensate(cfear_ctx* ctx, cfear_cloud* c, const double motion_xyt[3], int ccw) {
  if (!ctx || !c || !motion_xyt) return cfear_fail(ctx, CFEAR_ERR_INVALID, "compensate: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  // Affine3dToVectorXYeZ of the motion (utils.cpp:109-112): theta passes through atan2(sin, cos)
  const double th = atan2(sin(motion_xyt[2]), cos(motion_xyt[2]));
  hipLaunchKernelGGL(compensate_kernel, dim3(1), dim3(BLOCK_F), 0, ctx->stream, cloud_out(c), cloud_out(c), motion_xyt[0], motion_xyt[1], th, ccw);
  c->mirror_valid = c->h_block != nullptr;
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

int cfear_compensate_pair(cfear_ctx* ctx, cfear_cloud* c0, cfear_cloud* c1, const double motion_xyt[3], int ccw) {
  if (!ctx || !c0 || !c1 || c0 == c1 || !motion_xyt) return cfear_fail(ctx, CFEAR_ERR_INVALID, "compensate_pair: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const double th = atan2(sin(motion_xyt[2]), cos(motion_xyt[2]));
  hipLaunchKernelGGL(compensate_kernel, dim3(2), dim3(BLOCK_F), 0, ctx->stream, cloud_out(c0), cloud_out(c1), motion_xyt[0], motion_xyt[1], th, ccw);
  c0->mirror_valid = c0->h_block != nullptr; c1->mirror_valid = c1->h_block != nullptr;
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

// ---- scans -------------------------------------------------------------------------------------
}  // extern "C"
// cfear_tune VOXEL_ORDER = 1: the order PCL <= 1.9 leaves the points of a VoxelGrid voxel in (pointnormal.cpp:277-280 -> pcl::VoxelGrid::applyFilter).
// pcl/filters/impl/voxel_grid.hpp fills a vector of (voxel index, point index) in point order and calls std::sort on it with an operator<
// that compares the voxel index ONLY - an unstable sort, so the order of a voxel's points, and with it the last bit of its float centroid,
// is whatever libstdc++'s introsort leaves. Reproducing that needs the same call on the same sequence: the cloud comes to the host, the
// voxel indices are formed with the kernel's (= PCL's) float arithmetic, std::sort runs here, and the device gets every point's rank
// (features_dev.h: the sort key inside a voxel) and the point at every rank. Per-call scans only (a host round trip per scan: parity mode).
namespace {
struct cloud_point_index_idx {
  unsigned int idx, cloud_point_index;
  bool operator<(const cloud_point_index_idx& p) const { return idx < p.idx; }
};
int voxel_order_stdsort(cfear_ctx* ctx, const cfear_cloud* cloud, int cap_points, int** d_out /* [2][n]: rank, perm */) {
  *d_out = nullptr;
  int n = 0;
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(&n, cloud->d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  n = std::min(n, std::min(cloud->cap, cap_points));
  if (n <= 0) return CFEAR_OK;
  std::vector<float> xyi(3 * (size_t)n);
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(xyi.data(), cloud->d_xyi, sizeof(float) * xyi.size(), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const float leaf = (float)((double)(float)ctx->par.res / ctx->par.downsample_factor), inv = 1.0f / leaf;  // features_block
  float mnx = 3.4e38f, mxx = -3.4e38f, mny = 3.4e38f, mxy = -3.4e38f;
  for (int i = 0; i < n; i++) {
    const float x = xyi[3 * (size_t)i], y = xyi[3 * (size_t)i + 1];
    mnx = fminf(mnx, x); mxx = fmaxf(mxx, x); mny = fminf(mny, y); mxy = fmaxf(mxy, y);
  }
  const int min_b0 = (int)floorf(mnx * inv), max_b0 = (int)floorf(mxx * inv), min_b1 = (int)floorf(mny * inv);
  const int div0 = max_b0 - min_b0 + 1;
  std::vector<cloud_point_index_idx> v((size_t)n);
  for (int i = 0; i < n; i++) {
    const int ijk0 = (int)(floorf(xyi[3 * (size_t)i] * inv) - (float)min_b0), ijk1 = (int)(floorf(xyi[3 * (size_t)i + 1] * inv) - (float)min_b1);
    v[(size_t)i].idx = (unsigned int)(ijk0 + ijk1 * div0); v[(size_t)i].cloud_point_index = (unsigned int)i;
  }
  std::sort(v.begin(), v.end(), std::less<cloud_point_index_idx>());
  std::vector<int> rp(2 * (size_t)n);
  for (int r = 0; r < n; r++) { rp[(size_t)v[(size_t)r].cloud_point_index] = r; rp[(size_t)n + r] = (int)v[(size_t)r].cloud_point_index; }
  int* d = nullptr;
  if (hipMalloc(&d, sizeof(int) * rp.size()) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc voxel order");
  hipError_t e = hipMemcpyAsync(d, rp.data(), sizeof(int) * rp.size(), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // (rp is a local)
  if (e != hipSuccess) { (void)hipFree(d); return cfear_fail(ctx, CFEAR_ERR_HIP, "voxel order upload", e); }
  *d_out = d;
  return n;
}
}  // namespace
extern "C" {
int cfear_scan_create(cfear_ctx* ctx, const cfear_cloud* cloud, cfear_scan** scan) {
  if (!ctx || !cloud || !scan) return cfear_fail(ctx, CFEAR_ERR_INVALID, "scan_create: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  *scan = nullptr;
  int rc = set_kernel_attributes(ctx);
  if (rc != CFEAR_OK) return rc;
  const int cap = cloud->cap;
  if (cap >= (1 << 24)) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "scan_create: more than 2^24 points");
  rc = ensure_ctx_scratch(ctx, cap > ctx->A * ctx->par.k_strongest ? cap : ctx->A * ctx->par.k_strongest,
                          (MAX_SCANS - 1) * (cap > ctx->A * ctx->par.k_strongest ? cap : ctx->A * ctx->par.k_strongest));
  if (rc != CFEAR_OK) return rc;
  cfear_scan* s = new (std::nothrow) cfear_scan();
  if (!s) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "scan alloc");
  s->cap_points = cap; s->with_kd = ctx->tune_nn_tie == 2;
  const ScanLayout L = scan_layout(cap, cap, true, ctx->tune_nn_tie == 2);
  { void* blk = nullptr; rc = cfear_pool_alloc(ctx, L.total, &blk, &s->bytes); if (rc != CFEAR_OK) { delete s; return rc; } s->d_block = static_cast<unsigned char*>(blk); }
  const ScanDev h = scan_header(s->d_block, cap, cap, true, ctx->tune_nn_tie == 2);
  ScanDev back;
  int* d_vorder = nullptr;
  // the header goes in and comes back through the context's pinned staging (from / to the stack the two small copies are pageable: the runtime
  // stages and waits for each - on the per-sweep route two of the ~15 host round trips of a sweep)
  constexpr size_t HS = (sizeof(ScanDev) + 63) & ~(size_t)63;
  rc = cfear_ensure_hstage(ctx, 2 * HS);
  if (rc != CFEAR_OK) { cfear_pool_free(ctx, s->d_block, s->bytes); delete s; return rc; }
  memcpy(ctx->h_stage, &h, sizeof(h));
  hipError_t e = hipMemcpyAsync(s->d_block, ctx->h_stage, sizeof(h), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    const int capmax = cap > ctx->A * ctx->par.k_strongest ? cap : ctx->A * ctx->par.k_strongest;
    BlockScratch B = scratch_header(static_cast<unsigned char*>(ctx->d_scratch), capmax, (MAX_SCANS - 1) * capmax);
    if (ctx->tune_voxel_order == 1) {  // PCL <= 1.9's intra-voxel order (parity mode): ranks from a host std::sort
      const int nv = voxel_order_stdsort(ctx, cloud, cap, &d_vorder);
      if (nv < 0) { cfear_pool_free(ctx, s->d_block, s->bytes); delete s; return nv; }
      if (d_vorder) { B.vrank = d_vorder; B.vperm = d_vorder + nv; }
    }
    const FeatureParams P = feature_params(ctx);
    hipLaunchKernelGGL(features_kernel, dim3(1), dim3(BLOCK_F), 0, ctx->stream, reinterpret_cast<ScanDev*>(s->d_block), cloud->d_xyi,
                       cloud->d_n, P, B);
    e = hipGetLastError();
  }
  // the reference exits on an empty cloud (pointnormal.cpp:72-75); report it instead
  if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_stage + HS, s->d_block, sizeof(back), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) memcpy(&back, ctx->h_stage + HS, sizeof(back));
  if (d_vorder) (void)hipFree(d_vorder);
  if (e != hipSuccess) {
    cfear_pool_free(ctx, s->d_block, s->bytes);
    delete s;
    return cfear_fail(ctx, CFEAR_ERR_HIP, "scan_create", e);
  }
  if (back.status != 0) {
    cfear_pool_free(ctx, s->d_block, s->bytes);
    delete s;
    return cfear_fail(ctx, CFEAR_ERR_EMPTY, "scan_create: empty cloud (reference: 'error, cloud empty' + exit)");
  }
  s->n_cells = back.n_cells;
  *scan = s;
  return CFEAR_OK;
}

int cfear_scan_from_cells(cfear_ctx* ctx, const cfear_cell* cells, int n, cfear_scan** scan) {
  if (!ctx || !scan || n < 0 || (n > 0 && !cells)) return cfear_fail(ctx, CFEAR_ERR_INVALID, "scan_from_cells: bad argument");
  if (n >= (1 << 24)) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "scan_from_cells: more than 2^24 cells");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  *scan = nullptr;
  if (n == 0) return cfear_fail(ctx, CFEAR_ERR_EMPTY, "scan_from_cells: no cells (reference: 'error, cloud empty' + exit)");
  const int base = ctx->A * ctx->par.k_strongest, capmax = n > base ? n : base;
  int rc = ensure_ctx_scratch(ctx, capmax, (MAX_SCANS - 1) * capmax);
  if (rc != CFEAR_OK) return rc;
  cfear_scan* s = new (std::nothrow) cfear_scan();
  if (!s) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "scan alloc");
  s->cap_points = n; s->with_kd = ctx->tune_nn_tie == 2;
  const ScanLayout L = scan_layout(n, n, true, ctx->tune_nn_tie == 2);
  { void* blk = nullptr; rc = cfear_pool_alloc(ctx, L.total, &blk, &s->bytes); if (rc != CFEAR_OK) { delete s; return rc; } s->d_block = static_cast<unsigned char*>(blk); }
  const ScanDev h = scan_header(s->d_block, n, n, true, ctx->tune_nn_tie == 2);
  hipError_t e = hipMemcpyAsync(s->d_block, &h, sizeof(h), hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(s->d_block + L.cells, cells, sizeof(cfear_cell) * (size_t)n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    const BlockScratch B = scratch_header(static_cast<unsigned char*>(ctx->d_scratch), capmax, (MAX_SCANS - 1) * capmax);
    hipLaunchKernelGGL(scan_from_cells_kernel, dim3(1), dim3(BLOCK_F), 0, ctx->stream, reinterpret_cast<ScanDev*>(s->d_block), n, feature_params(ctx), B);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // h and the caller's cells are read until here
  if (e != hipSuccess) { cfear_pool_free(ctx, s->d_block, s->bytes); delete s; return cfear_fail(ctx, CFEAR_ERR_HIP, "scan_from_cells", e); }
  s->n_cells = n;
  *scan = s;
  return CFEAR_OK;
}

void cfear_scan_release(cfear_ctx* ctx, cfear_scan* s) {
  if (!s) return;
  cfear_pool_free(ctx, s->d_block, s->bytes);  // back to the context's pool (stream-ordered reuse, no synchronisation)
  delete s;
}

static int scan_header_download(cfear_ctx* ctx, const cfear_scan* s, ScanDev* h) {
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(h, s->d_block, sizeof(ScanDev), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

int cfear_scan_size(cfear_ctx* ctx, const cfear_scan* s, int* n_cells) {
  if (!ctx || !s || !n_cells) return cfear_fail(ctx, CFEAR_ERR_INVALID, "scan_size: bad argument");
  if (s->n_cells >= 0) { *n_cells = s->n_cells; return CFEAR_OK; }  // cfear_scan_create read the header back already
  ScanDev h;
  int rc = scan_header_download(ctx, s, &h);
  if (rc != CFEAR_OK) return rc;
  *n_cells = h.n_cells;
  return CFEAR_OK;
}

int cfear_scan_download_cells(cfear_ctx* ctx, const cfear_scan* s, cfear_cell* cells, int capacity, int* n) {
  if (!ctx || !s) return cfear_fail(ctx, CFEAR_ERR_INVALID, "scan_download_cells: bad argument");
  ScanDev h;
  int rc = scan_header_download(ctx, s, &h);
  if (rc != CFEAR_OK) return rc;
  if (n) *n = h.n_cells;
  const int cnt = h.n_cells < capacity ? h.n_cells : capacity;
  if (cnt > 0 && cells) {
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(cells, h.cells, sizeof(cfear_cell) * (size_t)cnt, hipMemcpyDeviceToHost, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return CFEAR_OK;
}

int cfear_scan_closest(cfear_ctx* ctx, const cfear_scan* s, const double* qxy, int nq, double d, int32_t* idx) {
  if (!ctx || !s || !qxy || !idx || nq <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "scan_closest: bad argument");
  { cfear_scan* one = const_cast<cfear_scan*>(s); const int trc = check_tie_rule_scans(ctx, &one, 1, "scan_closest"); if (trc != CFEAR_OK) return trc; }
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  double* dq = nullptr; int* di = nullptr;
  if (hipMalloc(&dq, sizeof(double) * 2 * (size_t)nq) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc queries");
  if (hipMalloc(&di, sizeof(int) * (size_t)nq) != hipSuccess) { (void)hipFree(dq); return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc idx"); }
  hipError_t e = hipMemcpyAsync(dq, qxy, sizeof(double) * 2 * (size_t)nq, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(closest_kernel, dim3((nq + 255) / 256), dim3(256), 0, ctx->stream, reinterpret_cast<const ScanDev*>(s->d_block), dq, nq, d, di, ctx->tune_nn_tie);
    e = hipMemcpyAsync(idx, di, sizeof(int) * (size_t)nq, hipMemcpyDeviceToHost, ctx->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(dq); (void)hipFree(di);
  if (e != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_HIP, "scan_closest", e);
  return CFEAR_OK;
}

// ---- registration ------------------------------------------------------------------------------
static int register_impl(cfear_ctx* ctx, cfear_scan* const* scans, int n, double* poses_xyt, const double* prior_cov6, double* cov6_last,
                         cfear_reg_summary* summary) {
  if (!ctx || !scans || !poses_xyt || n < 2) return cfear_fail(ctx, CFEAR_ERR_INVALID, "register: need >= 2 scans and poses");
  if (n > MAX_SCANS) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "register: more than 64 scans");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int rc = set_kernel_attributes(ctx);
  if (rc != CFEAR_OK) return rc;
  int capmax = ctx->A * ctx->par.k_strongest;
  for (int i = 0; i < n; i++) {
    if (!scans[i]) return cfear_fail(ctx, CFEAR_ERR_INVALID, "register: null scan");
    if (scans[i]->cap_points > capmax) capmax = scans[i]->cap_points;
  }
  if ((rc = check_tie_rule_scans(ctx, scans, n, "register")) != CFEAR_OK) return rc;
  rc = ensure_ctx_scratch(ctx, capmax, (MAX_SCANS - 1) * capmax);
  if (rc != CFEAR_OK) return rc;
  const ScratchLayout L = scratch_layout(capmax, (MAX_SCANS - 1) * capmax);
  unsigned char* base = static_cast<unsigned char*>(ctx->d_scratch);
  const BlockScratch B = scratch_header(base, capmax, (MAX_SCANS - 1) * capmax);
  unsigned char* tail = base + L.total;
  double* d_poses = reinterpret_cast<double*>(tail);
  double* d_cov = d_poses + 3 * MAX_SCANS;
  ScanDev** d_ptrs = reinterpret_cast<ScanDev**>(d_cov + 36);
  cfear_reg_summary* d_sum = reinterpret_cast<cfear_reg_summary*>(reinterpret_cast<unsigned char*>(d_ptrs) + sizeof(void*) * MAX_SCANS);
  // arguments and results travel through pinned staging: one copy each way and one synchronisation per registration (round 6; it was
  // three small pageable copies in, three out). Device layout of the tail: poses | cov | scan pointers | summary | prior.
  const size_t in_bytes = sizeof(double) * (3 * MAX_SCANS + 36) + sizeof(void*) * MAX_SCANS;
  const size_t sum_bytes = ((sizeof(cfear_reg_summary) + 15) / 16) * 16;
  rc = cfear_ensure_hstage(ctx, in_bytes + sum_bytes + sizeof(double) * 36);
  if (rc != CFEAR_OK) return rc;
  double* h_poses = reinterpret_cast<double*>(ctx->h_stage);
  double* h_cov = h_poses + 3 * MAX_SCANS;
  ScanDev** h_ptrs = reinterpret_cast<ScanDev**>(h_cov + 36);
  memset(ctx->h_stage, 0, in_bytes);
  for (int i = 0; i < n; i++) h_ptrs[i] = reinterpret_cast<ScanDev*>(scans[i]->d_block);
  memcpy(h_poses, poses_xyt, sizeof(double) * 3 * n);
  // a registration that fails does not touch the caller's covariance (reg_cov of n_scan_normal_reg::Register, n_scan_normal.cpp:82-187):
  // the device copy starts as the caller's matrix, so that what comes back is the caller's matrix (not the previous call's result)
  if (cov6_last) memcpy(h_cov, cov6_last, sizeof(double) * 36);
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_poses, ctx->h_stage, in_bytes, hipMemcpyHostToDevice, ctx->stream));
  const RegParams P = reg_params(ctx);
  double* d_prior = nullptr;
  if (prior_cov6) {  // staged behind the summary, in the tail of the context scratch
    d_prior = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(d_sum) + sum_bytes);
    double* h_prior = reinterpret_cast<double*>(ctx->h_stage + in_bytes + sum_bytes);
    memcpy(h_prior, prior_cov6, sizeof(double) * 36);
    CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(d_prior, h_prior, sizeof(double) * 36, hipMemcpyHostToDevice, ctx->stream));
  }
  hipLaunchKernelGGL(register_kernel, dim3(1), dim3(BLOCK_R), 0, ctx->stream, d_ptrs, n, d_poses, d_cov, P, B, d_sum, d_prior);
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_stage, d_poses, in_bytes + sizeof(cfear_reg_summary), hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(poses_xyt, h_poses, sizeof(double) * 3 * n);
  if (cov6_last) memcpy(cov6_last, h_cov, sizeof(double) * 36);
  if (summary) memcpy(summary, ctx->h_stage + in_bytes, sizeof(cfear_reg_summary));
  if (reinterpret_cast<const cfear_reg_summary*>(ctx->h_stage + in_bytes)->assoc_path < 0)  // (registration_dev.h: the kd descent's stack did not fit the match scratch)
    return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "register: NN_TIE_RULE could not be honoured for this problem size (match scratch too small for the kd-tree descent); the result used the production rule");
  return CFEAR_OK;
}

int cfear_register(cfear_ctx* ctx, cfear_scan* const* scans, int n, double* poses_xyt, double* cov6_last,
                   cfear_reg_summary* summary) {
  return register_impl(ctx, scans, n, poses_xyt, nullptr, cov6_last, summary);
}

int cfear_register_soft(cfear_ctx* ctx, cfear_scan* const* scans, int n, double* poses_xyt, const double* prior_cov6, double* cov6_last,
                        cfear_reg_summary* summary) {
  if (!prior_cov6) return cfear_fail(ctx, CFEAR_ERR_INVALID, "register_soft: null prior covariance");
  return register_impl(ctx, scans, n, poses_xyt, prior_cov6, cov6_last, summary);
}

int cfear_get_cost(cfear_ctx* ctx, cfear_scan* const* scans, int n, const double* poses_xyt, int itr, double* score, double* residuals,
                   int capacity, int* n_residuals) {
  if (!ctx || !scans || !poses_xyt || !score || !n_residuals || n < 2 || capacity < 0 || (capacity > 0 && !residuals))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "get_cost: need >= 2 scans, poses and output pointers");
  if (n > MAX_SCANS) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "get_cost: more than 64 scans");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int capmax = ctx->A * ctx->par.k_strongest;
  for (int i = 0; i < n; i++) {
    if (!scans[i]) return cfear_fail(ctx, CFEAR_ERR_INVALID, "get_cost: null scan");
    if (scans[i]->cap_points > capmax) capmax = scans[i]->cap_points;
  }
  int rc = check_tie_rule_scans(ctx, scans, n, "get_cost");
  if (rc != CFEAR_OK) return rc;
  rc = ensure_ctx_scratch(ctx, capmax, (MAX_SCANS - 1) * capmax);
  if (rc != CFEAR_OK) return rc;
  const ScratchLayout L = scratch_layout(capmax, (MAX_SCANS - 1) * capmax);
  unsigned char* base = static_cast<unsigned char*>(ctx->d_scratch);
  const BlockScratch B = scratch_header(base, capmax, (MAX_SCANS - 1) * capmax);
  unsigned char* tail = base + L.total;
  double* d_poses = reinterpret_cast<double*>(tail);
  double* d_score = d_poses + 3 * MAX_SCANS;  // the covariance slot of cfear_register
  int* d_nres = reinterpret_cast<int*>(d_score + 1);
  ScanDev** d_ptrs = reinterpret_cast<ScanDev**>(d_score + 36);
  double* d_res = nullptr;
  if (capacity > 0 && hipMalloc(&d_res, sizeof(double) * (size_t)capacity) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc residuals");
  ScanDev* h_ptrs[MAX_SCANS];
  for (int i = 0; i < n; i++) h_ptrs[i] = reinterpret_cast<ScanDev*>(scans[i]->d_block);
  hipError_t e = hipMemcpyAsync(d_ptrs, h_ptrs, sizeof(void*) * n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_poses, poses_xyt, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(get_cost_kernel, dim3(1), dim3(BLOCK_R), 0, ctx->stream, d_ptrs, n, d_poses, reg_params(ctx), B, itr, d_score, d_res,
                       capacity, d_nres);
    e = hipGetLastError();
  }
  int nres = -1;
  if (e == hipSuccess) e = hipMemcpyAsync(score, d_score, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(&nres, d_nres, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess && nres > 0 && capacity > 0)
    e = hipMemcpy(residuals, d_res, sizeof(double) * (size_t)(nres < capacity ? nres : capacity), hipMemcpyDeviceToHost);
  if (d_res) (void)hipFree(d_res);
  if (e != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_HIP, "get_cost", e);
  *n_residuals = nres;
  if (nres < 0) return cfear_fail(ctx, CFEAR_ERR_EMPTY, "get_cost: too few residuals");  // GetCost returns false (:205-208)
  return CFEAR_OK;
}

// ---- cost-sampling covariance (odometrykeyframefuser.cpp:261-380) ------------------------------------
namespace {
// Minimum-norm least squares of A c = b (A: m x 10), what Eigen's bdcSvd().solve() returns (odometrykeyframefuser.cpp:337), by a
// one-sided Jacobi (Hestenes) singular value decomposition: plane rotations from the right make the columns of W = A V
// mutually orthogonal; then the singular values are the column norms, U = W / sigma, and c = V diag(1 / sigma) U^T b over the
// singular values above the rank threshold (Eigen's SVDBase::threshold(): diagSize = min(m, n) times epsilon, times sigma_max; numpy's lstsq
// default is max(m, n) - with 27-125 samples 3-12 x higher, which would cut a weakly observed yaw direction Eigen keeps). Works on A itself - no normal
// equations - so nothing is lost to squaring the condition number (the yaw column is ~1e-5 of the others).
static void lstsq10_svd(int m, const double* A, const double* b, double c[10]) {
  const int n = 10;
  std::vector<double> W((size_t)m * n);
  double V[100];
  for (int i = 0; i < m * n; i++) W[i] = A[i];
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < m; i++) { const double x = W[(size_t)i * n + p], y = W[(size_t)i * n + q]; al += x * x; be += y * y; ga += x * y; }
        if (ga == 0.0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
        rotated = true;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
        for (int i = 0; i < m; i++) {
          const double x = W[(size_t)i * n + p], y = W[(size_t)i * n + q];
          W[(size_t)i * n + p] = cs * x - sn * y; W[(size_t)i * n + q] = sn * x + cs * y;
        }
        for (int i = 0; i < n; i++) {
          const double x = V[i * n + p], y = V[i * n + q];
          V[i * n + p] = cs * x - sn * y; V[i * n + q] = sn * x + cs * y;
        }
      }
    if (!rotated) break;
  }
  double sig[10], smax = 0;
  for (int j = 0; j < n; j++) {
    double q = 0;
    for (int i = 0; i < m; i++) q += W[(size_t)i * n + j] * W[(size_t)i * n + j];
    sig[j] = sqrt(q);
    if (sig[j] > smax) smax = sig[j];
  }
  const double thr = (double)(m < n ? m : n) * 2.220446049250313e-16 * smax;
  for (int k = 0; k < n; k++) c[k] = 0.0;
  for (int j = 0; j < n; j++) {
    if (!(sig[j] > thr)) continue;
    double q = 0;
    for (int i = 0; i < m; i++) q += W[(size_t)i * n + j] * b[i];  // sigma_j * (u_j . b)
    q /= sig[j] * sig[j];
    for (int k = 0; k < n; k++) c[k] += V[k * n + j] * q;
  }
}
static void linspace(double start, double end, int num, std::vector<double>& v) {  // odometrykeyframefuser.cpp:497-524
  v.clear();
  if (num <= 0) return;
  if (num == 1) { v.push_back(start); return; }
  const double delta = (end - start) / ((double)num - 1);
  for (int i = 0; i < num - 1; i++) v.push_back(start + delta * i);
  v.push_back(end);
}
}  // namespace

int cfear_cov_by_sampling(cfear_ctx* ctx, cfear_scan* const* scans, int n, const double* poses_xyt, int itr, double xy_range,
                          double yaw_range, int samples_per_axis, double covariance_scaler, double final_cost, int num_residuals,
                          double* cov6, int* success, double* sample_costs) {
  if (!ctx || !scans || !poses_xyt || !cov6 || !success || n < 2 || samples_per_axis < 1 || samples_per_axis > 32)
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "cov_by_sampling: bad argument");
  if (n > MAX_SCANS) return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "cov_by_sampling: more than 64 scans");
  *success = 0;
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i < n; i++) if (!scans[i]) return cfear_fail(ctx, CFEAR_ERR_INVALID, "cov_by_sampling: null scan");
  { const int trc = check_tie_rule_scans(ctx, scans, n, "cov_by_sampling"); if (trc != CFEAR_OK) return trc; }
  int nsrc = 0;
  int rc = cfear_scan_size(ctx, scans[n - 1], &nsrc);
  if (rc != CFEAR_OK) return rc;
  const int steps = samples_per_axis, m = steps * steps * steps, L = 3 * (n - 1);
  // match scratch per sample: a pair of capacity per (keyframe, source cell); under a non-production tie rule the kd descent's per-thread stacks
  // live there too (registration_dev.h associate_pair_rule: 64 B per pair of capacity for blockDim x 64 entries of 16 B)
  const int cap = std::max((n - 1) * (nsrc > 0 ? nsrc : 1), ctx->tune_nn_tie != 0 ? 4096 : 1);
  std::vector<double> xs, ths;  // :277-290
  linspace(-xy_range * 0.5, xy_range * 0.5, steps, xs);
  linspace(-yaw_range * 0.5, yaw_range * 0.5, steps, ths);
  std::vector<double> samples(3 * (size_t)m), A(10 * (size_t)m), costs((size_t)m);
  std::vector<int> nres((size_t)m);
  int k = 0;
  for (int it = 0; it < steps; it++)      // the reference's loop order (:294-296)
    for (int ix = 0; ix < steps; ix++)
      for (int iy = 0; iy < steps; iy++, k++) {
        samples[3 * k] = xs[ix] + poses_xyt[L]; samples[3 * k + 1] = xs[iy] + poses_xyt[L + 1]; samples[3 * k + 2] = ths[it] + poses_xyt[L + 2];
        const double x = xs[ix], y = xs[iy], z = ths[it];
        double* r = &A[10 * (size_t)k];
        r[0] = x * x; r[1] = y * y; r[2] = z * z; r[3] = x * y; r[4] = y * z; r[5] = z * x; r[6] = x; r[7] = y; r[8] = z; r[9] = 1.0;  // :325-336
      }
  // device buffers: poses, samples, scan pointers, costs, residual counts, per-sample match arrays
  const size_t bytes = sizeof(double) * (3 * (size_t)n + 3 * (size_t)m + (size_t)m) + sizeof(void*) * (size_t)n + sizeof(int) * (size_t)m +
                       (sizeof(double) * 8 + 3 * sizeof(int)) * (size_t)m * cap + 256;  // (three ints of association scratch per pair: the grouped path, as scratch_layout)
  unsigned char* d = nullptr;
  if (hipMalloc(&d, bytes) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc cost samples");
  double* d_match = reinterpret_cast<double*>(d);
  double* d_poses = d_match + 8 * (size_t)m * cap;
  double* d_samples = d_poses + 3 * n;
  double* d_costs = d_samples + 3 * m;
  ScanDev** d_ptrs = reinterpret_cast<ScanDev**>(d_costs + m);
  int* d_nres = reinterpret_cast<int*>(d_ptrs + n);
  int* d_assoc = d_nres + m;
  ScanDev* h_ptrs[MAX_SCANS];
  for (int i = 0; i < n; i++) h_ptrs[i] = reinterpret_cast<ScanDev*>(scans[i]->d_block);
  hipError_t e = hipMemcpyAsync(d_ptrs, h_ptrs, sizeof(void*) * n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_poses, poses_xyt, sizeof(double) * 3 * n, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_samples, samples.data(), sizeof(double) * 3 * m, hipMemcpyHostToDevice, ctx->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(get_cost_samples_kernel, dim3(m), dim3(BLOCK_R), 0, ctx->stream, d_ptrs, n, d_poses, d_samples, reg_params(ctx), d_match,
                       d_assoc, cap, itr, d_costs, d_nres);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(costs.data(), d_costs, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(nres.data(), d_nres, sizeof(int) * m, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  if (e != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_HIP, "cov_by_sampling", e);
  double last = 0.0;  // a failed GetCost leaves sample_cost at its previous value (:305: the return value is ignored)
  for (int i = 0; i < m; i++) { if (nres[i] >= 0) last = costs[i]; costs[i] = last; }
  if (sample_costs) memcpy(sample_costs, costs.data(), sizeof(double) * m);
  double c[10];
  lstsq10_svd(m, A.data(), costs.data(), c);
  const double H[9] = {2 * c[0], c[3], c[5], c[3], 2 * c[1], c[4], c[5], c[4], 2 * c[2]};  // :340-343
  // "all eigenvalues positive" (:355-358) of a symmetric matrix = positive definite = all leading principal minors positive
  // (Sylvester); the inverse by cofactors, as Eigen's Matrix3d::inverse() (:363)
  const double C00 = H[4] * H[8] - H[5] * H[7], C01 = H[5] * H[6] - H[3] * H[8], C02 = H[3] * H[7] - H[4] * H[6];
  const double det = H[0] * C00 + H[1] * C01 + H[2] * C02;
  const double minor2 = H[0] * H[4] - H[1] * H[3];
  if (!(H[0] > 0.0 && minor2 > 0.0 && det > 0.0)) return CFEAR_OK;  // not convex: sampling not used for this scan
  if (num_residuals - 3 == 0) return CFEAR_OK;                      // GetCovarianceScaler false (n_scan_normal.cpp:435-441)
  const double score_scale = final_cost / (double)(num_residuals - 3);
  const double id = 1.0 / det;
  const double Hi[9] = {C00 * id, (H[2] * H[7] - H[1] * H[8]) * id, (H[1] * H[5] - H[2] * H[4]) * id,
                        C01 * id, (H[0] * H[8] - H[2] * H[6]) * id, (H[2] * H[3] - H[0] * H[5]) * id,
                        C02 * id, (H[1] * H[6] - H[0] * H[7]) * id, minor2 * id};
  double C3[9];
  for (int i = 0; i < 9; i++) C3[i] = 2.0 * Hi[i] * score_scale * covariance_scaler;  // :363
  for (int i = 0; i < 36; i++) cov6[i] = (i % 7 == 0) ? 1.0 : 0.0;  // :366-373
  cov6[0] = C3[0]; cov6[1] = C3[1]; cov6[6] = C3[3]; cov6[7] = C3[4];
  cov6[35] = C3[8]; cov6[5] = C3[2]; cov6[11] = C3[5]; cov6[30] = C3[6]; cov6[31] = C3[7];
  *success = 1;
  return CFEAR_OK;
}

// ---- batched odometry --------------------------------------------------------------------------
void cfear_odometry_destroy(cfear_ctx* ctx, cfear_odometry* o) {
  if (!o) return;
  if (ctx) {
    (void)hipSetDevice(ctx->device);
    std::vector<hipStream_t> all = o->so;
    all.push_back(o->sf);
    for (hipStream_t st : all) {
      if (!st) continue;
      (void)hipStreamSynchronize(st);
      for (size_t i = 0; i < ctx->aux_streams.size(); i++)
        if (ctx->aux_streams[i] == st) { ctx->aux_streams.erase(ctx->aux_streams.begin() + i); break; }
    }
    (void)hipStreamSynchronize(ctx->stream);
  }
  if (ctx && o->rp_stream) {
    (void)hipStreamSynchronize(o->rp_stream);
    for (size_t i = 0; i < ctx->aux_streams.size(); i++)
      if (ctx->aux_streams[i] == o->rp_stream) { ctx->aux_streams.erase(ctx->aux_streams.begin() + i); break; }
    (void)hipStreamSynchronize(ctx->stream);
  }
  void* ptrs[] = {o->d_scans, o->d_scan_ptrs, o->d_scratch, o->d_scratch_hdr, o->d_states, o->d_poses_work, o->d_cov_work,
                  o->d_summaries, o->d_poses_out, o->d_slots[0], o->d_slots[1], o->d_polar, o->d_phase_times,
                  o->rp_polar[0], o->rp_polar[1], o->rp_slots[0], o->rp_slots[1], o->d_records, o->d_flags, o->d_order, o->d_work,
                  o->d_cloud, o->d_cloud_n, o->d_cfar_rows, o->rp_cloud[0], o->rp_cloud[1], o->rp_cloud_n[0], o->rp_cloud_n[1], o->rp_cfar_rows};
  for (hipEvent_t e : {o->rp_filt[0], o->rp_filt[1], o->rp_used[0], o->rp_used[1], o->rp_in}) if (e) (void)hipEventDestroy(e);
  if (o->rp_stream) (void)hipStreamDestroy(o->rp_stream);
  for (void* p : ptrs) if (p) (void)hipFree(p);
  for (hipEvent_t e : o->pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : {o->ev_in, o->ev_copied, o->ev_filt[0], o->ev_filt[1]}) if (e) (void)hipEventDestroy(e);
  for (auto* v : {&o->ev_free[0], &o->ev_free[1], &o->ev_done}) for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
  if (o->sf) (void)hipStreamDestroy(o->sf);
  for (hipStream_t st : o->so) if (st) (void)hipStreamDestroy(st);
  delete o;
}

int cfear_odometry_reset(cfear_ctx* ctx, cfear_odometry* o) {
  if (!ctx || !o) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_reset: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  std::vector<SeqState> st((size_t)o->B);
  for (auto& s : st) {
    memset(&s, 0, sizeof(s));
    s.T_prev.l0 = s.T_prev.l3 = 1; s.Tmot.l0 = s.Tmot.l3 = 1; s.Tcurrent.l0 = s.Tcurrent.l3 = 1;  // odometrykeyframefuser.cpp:34-38
  }
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(o->d_states, st.data(), sizeof(SeqState) * st.size(), hipMemcpyHostToDevice, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipMemsetAsync(o->d_summaries, 0, sizeof(cfear_reg_summary) * (size_t)o->B, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipMemsetAsync(o->d_poses_out, 0, sizeof(double) * 3 * (size_t)o->B, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipMemsetAsync(o->d_cov_work, 0, sizeof(double) * 36 * (size_t)o->B, ctx->stream));
  if (o->d_flags) CFEAR_HIP_CHECK(ctx, hipMemsetAsync(o->d_flags, 0, sizeof(int) * ((size_t)o->B + 1), ctx->stream));
  o->order_ready = false;
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return CFEAR_OK;
}

int cfear_odometry_create(cfear_ctx* ctx, int n_sequences, cfear_odometry** out) {
  if (!ctx || !out || n_sequences <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_create: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  *out = nullptr;
  int rc = set_kernel_attributes(ctx);
  if (rc != CFEAR_OK) return rc;
  cfear_odometry* o = new (std::nothrow) cfear_odometry();
  if (!o) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "odometry alloc");
  if (ctx->tune_voxel_order != 0) {
    delete o;
    return cfear_fail(ctx, CFEAR_ERR_UNSUPPORTED, "odometry_create: cfear_tune VOXEL_ORDER = 1 (PCL <= 1.9's std::sort order) needs a host round trip per scan: per-call scans only "
                      "(cfear_scan_create; the mirror classes of cfear_host.hpp use it)");
  }
  const int B = n_sequences, s = ctx->par.submap_scan_size;
  o->filter = ctx->par.filter_type;
  o->large_kernel = ctx->tune_large_kernel;
  { int ncu = 0; if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && ncu > 0) o->n_cus = ncu; }
  o->B = B; o->nslots = s + 1; o->cap_points = o->filter == CFEAR_FILTER_CACFAR ? odo_cfar_points(ctx) : ctx->A * ctx->par.k_strongest;
  // cells per scan the blocks are sized for: cfear_tune MAX_CELLS; by default every filtered point (cannot overflow) - except for a submap of
  // more than seven keyframes, where that default would be hundreds of MB per sequence (280 MB at submap_scan_size 50, k 40) for scans of a
  // few hundred to ~1500 cells: 4096 then (launch/oxford_demo:62-71 works without a tune call; an overflow is loud, CFEAR_ERR_CAPACITY)
  o->cap_cells = ctx->tune_max_cells > 0 ? std::min(ctx->tune_max_cells, o->cap_points) : (s > 7 ? std::min(o->cap_points, 4096) : o->cap_points);
  // residual blocks of a registration <= keyframes x cells of the current scan (one match per source cell and keyframe,
  // n_scan_normal.cpp:242,258); the association parks four results per source cell in the same scratch
  o->pair_cap = std::max(s, 4) * o->cap_cells;
  if (ctx->tune_nn_tie != 0) o->pair_cap = std::max(o->pair_cap, 8192);  // room for the kd descent's per-thread stacks (associate_pair_rule, 512-thread kernels)
  o->with_kd = ctx->tune_nn_tie == 2;
  const ScanLayout SL = scan_layout(o->cap_points, o->cap_cells, false, o->with_kd);
  const ScratchLayout WL = scratch_layout(o->cap_points, o->pair_cap);
  {  // refuse what cannot fit with a message that names the numbers (a bare NOMEM after gigabytes of partial allocations helps nobody)
    size_t free_b = 0, total_b = 0;
    const size_t per_seq = SL.total * (size_t)o->nslots + WL.total + sizeof(uint32_t) * 2 * (size_t)o->cap_points + sizeof(SeqState) + 4096;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && per_seq * (size_t)B > free_b) {
      char msg[512];
      snprintf(msg, sizeof(msg), "odometry_create: %d sequences x %.1f MB (%d scan slots of %.2f MB for %d points / %d cells each + %.1f MB of scratch for %d residual "
               "blocks) = %.1f GB, %.1f GB free: at most %zu sequences fit - or size the scans for fewer cells (cfear_tune CFEAR_TUNE_MAX_CELLS, now %d)",
               B, per_seq / 1048576.0, o->nslots, SL.total / 1048576.0, o->cap_points, o->cap_cells, WL.total / 1048576.0, o->pair_cap,
               per_seq * (double)B / 1073741824.0, free_b / 1073741824.0, free_b / per_seq, o->cap_cells);
      delete o;
      return cfear_fail(ctx, CFEAR_ERR_NOMEM, msg);
    }
  }
  bool ok = true;
  ok = ok && hipMalloc(&o->d_scans, SL.total * (size_t)B * o->nslots) == hipSuccess;
  ok = ok && hipMalloc(&o->d_scan_ptrs, sizeof(ScanDev*) * (size_t)B * o->nslots) == hipSuccess;
  ok = ok && hipMalloc(&o->d_scratch, WL.total * (size_t)B) == hipSuccess;
  ok = ok && hipMemset(o->d_scratch, 0, WL.total * (size_t)B) == hipSuccess;  // dense voxel tables start all-zero
  ok = ok && hipMalloc(&o->d_scratch_hdr, sizeof(BlockScratch) * (size_t)B) == hipSuccess;
  ok = ok && hipMalloc(&o->d_states, sizeof(SeqState) * (size_t)B) == hipSuccess;
  ok = ok && hipMalloc(&o->d_poses_work, sizeof(double) * 3 * MAX_SCANS * (size_t)B) == hipSuccess;
  ok = ok && hipMalloc(&o->d_cov_work, sizeof(double) * 36 * (size_t)B) == hipSuccess;
  ok = ok && hipMalloc(&o->d_summaries, sizeof(cfear_reg_summary) * (size_t)B) == hipSuccess;
  ok = ok && hipMalloc(&o->d_poses_out, sizeof(double) * 3 * (size_t)B) == hipSuccess;
  if (o->filter == CFEAR_FILTER_CACFAR) {
    ok = ok && hipMalloc(&o->d_cloud, sizeof(float) * 3 * (size_t)B * o->cap_points) == hipSuccess;
    ok = ok && hipMalloc(&o->d_cloud_n, sizeof(int) * (size_t)B) == hipSuccess;
    ok = ok && hipMalloc(&o->d_cfar_rows, sizeof(int) * cfear_cfar_scratch_ints(ctx, (size_t)B)) == hipSuccess;
  } else {
    ok = ok && hipMalloc(&o->d_slots[0], sizeof(uint32_t) * (size_t)B * o->cap_points) == hipSuccess;
    ok = ok && hipMalloc(&o->d_slots[1], sizeof(uint32_t) * (size_t)B * o->cap_points) == hipSuccess;
  }
  // (a sweep's CA-CFAR detections may exceed the points the object holds; cfear_odometry_step_cloud_device allocates the word when first used)
  if (ok && (o->cap_cells < o->cap_points || o->filter == CFEAR_FILTER_CACFAR)) ok = hipMalloc(&o->d_flags, sizeof(int) * ((size_t)B + 1)) == hipSuccess && hipMemset(o->d_flags, 0, sizeof(int) * ((size_t)B + 1)) == hipSuccess;
  if (ok && ctx->tune_reg_order && B >= 2)
    ok = hipMalloc(&o->d_order, sizeof(int) * (size_t)B) == hipSuccess && hipMalloc(&o->d_work, sizeof(unsigned) * (size_t)B) == hipSuccess &&
         hipMemset(o->d_work, 0, sizeof(unsigned) * (size_t)B) == hipSuccess;
  if (!ok) { cfear_odometry_destroy(ctx, o); return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc odometry state"); }
  o->scan_stride = SL.total;
  std::vector<ScanDev*> ptrs((size_t)B * o->nslots);
  std::vector<BlockScratch> hdrs((size_t)B);
  std::vector<ScanDev> scan_hdrs((size_t)B * o->nslots);  // all headers built on the host, uploaded with one strided copy
  for (int q = 0; q < B; q++) {
    for (int j = 0; j < o->nslots; j++) {
      unsigned char* blk = o->d_scans + SL.total * ((size_t)q * o->nslots + j);
      scan_hdrs[(size_t)q * o->nslots + j] = scan_header(blk, o->cap_points, o->cap_cells, false, o->with_kd);
      ptrs[(size_t)q * o->nslots + j] = reinterpret_cast<ScanDev*>(blk);
    }
    hdrs[q] = scratch_header(o->d_scratch + WL.total * (size_t)q, o->cap_points, o->pair_cap);
  }
  ok = ok && hipMemcpy2D(o->d_scans, SL.total, scan_hdrs.data(), sizeof(ScanDev), sizeof(ScanDev), scan_hdrs.size(), hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && hipMemcpy(o->d_scan_ptrs, ptrs.data(), sizeof(ScanDev*) * ptrs.size(), hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && hipMemcpy(o->d_scratch_hdr, hdrs.data(), sizeof(BlockScratch) * hdrs.size(), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { cfear_odometry_destroy(ctx, o); return cfear_fail(ctx, CFEAR_ERR_HIP, "odometry state upload"); }
  rc = cfear_odometry_reset(ctx, o);
  if (rc != CFEAR_OK) { cfear_odometry_destroy(ctx, o); return rc; }
  ok = hipEventCreateWithFlags(&o->ev_copied, hipEventDisableTiming) == hipSuccess;
  o->overlap = ctx->tune_odo_overlap < 0 ? 0 : (ctx->tune_odo_overlap > 8 ? 8 : ctx->tune_odo_overlap);
  if (o->overlap > B) o->overlap = B;
  if (o->filter == CFEAR_FILTER_CACFAR) o->overlap = 0;  // (the filter-ahead streams are the k-strongest filter's)
  if (ok && o->overlap) {
    int least = 0, greatest = 0;  // numerically lower = higher priority
    ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess;
    // CFEAR_TUNE_FILTER_CUS = F: the filter gets F compute units of its own, the odometry streams the others: the HBM-bound
    // kernel and the latency-bound ones stop competing for a unit's registers and LDS. What the mask bits select was measured
    // (tools/cu_mask_probe.py): the unit of masking on this GPU is a group of 8 consecutive bits - any bit set enables the
    // whole group, 32 groups in all - so F is rounded to groups, and the groups are taken in a spread order (0, 4, 8, ...,
    // then 2, 6, ..., then the odd ones) so that every quarter of the bit range contributes alike.
    std::vector<uint32_t> mask_f, mask_o;
    int ncu = 0;
    if (ctx->tune_filter_cus > 0 && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && ncu >= 16) {
      const int ngroups = ncu / 8;
      const int gf = std::max(1, std::min(ngroups - 1, (ctx->tune_filter_cus + 4) / 8));
      std::vector<int> order;
      for (int phase : {0, 2, 1, 3}) for (int g = phase; g < ngroups; g += 4) order.push_back(g);
      mask_f.assign((size_t)(ncu + 31) / 32, 0u); mask_o.assign((size_t)(ncu + 31) / 32, 0u);
      for (int r = 0; r < ngroups; r++) {
        std::vector<uint32_t>& m = r < gf ? mask_f : mask_o;
        for (int b = 0; b < 8; b++) { const int i = order[(size_t)r] * 8 + b; m[(size_t)i / 32] |= 1u << (i % 32); }
      }
      for (int i = ngroups * 8; i < ncu; i++) mask_o[(size_t)i / 32] |= 1u << (i % 32);
    }
    if (!mask_f.empty()) ok = ok && hipExtStreamCreateWithCUMask(&o->sf, (uint32_t)mask_f.size(), mask_f.data()) == hipSuccess;
    else ok = ok && hipStreamCreateWithPriority(&o->sf, hipStreamNonBlocking, least) == hipSuccess;
    for (int i = 0; ok && i < o->overlap; i++) {
      hipStream_t st = nullptr;
      if (!mask_o.empty()) ok = hipExtStreamCreateWithCUMask(&st, (uint32_t)mask_o.size(), mask_o.data()) == hipSuccess;
      else ok = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest) == hipSuccess;
      if (ok) o->so.push_back(st);
      for (auto* v : {&o->ev_free[0], &o->ev_free[1], &o->ev_done}) {
        hipEvent_t e = nullptr;
        ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
        if (ok) v->push_back(e);
      }
    }
    for (hipEvent_t* e : {&o->ev_in, &o->ev_filt[0], &o->ev_filt[1]})
      ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  }
  if (!ok) { cfear_odometry_destroy(ctx, o); return cfear_fail(ctx, CFEAR_ERR_HIP, "odometry stream creation"); }
  if (o->overlap) { ctx->aux_streams.push_back(o->sf); for (hipStream_t st : o->so) ctx->aux_streams.push_back(st); }
  *out = o;
  return CFEAR_OK;
}

// one sweep of every sequence from clouds on the device, on the context stream
static int odo_step_clouds(cfear_ctx* ctx, cfear_odometry* o, const float* d_xyi, int capacity, const int* d_counts, bool filter_events) {
  const OdoParams OP = odo_params(ctx, o);
  int rc = CFEAR_OK;
  if (o->profile && !filter_events) {  // (keeps filter_events / stage_events paired per step for the profile readers)
    if ((rc = odo_timed_event(ctx, o, o->filter_events, ctx->stream)) != CFEAR_OK) return rc;
    if ((rc = odo_timed_event(ctx, o, o->filter_events, ctx->stream)) != CFEAR_OK) return rc;
  }
  if (o->profile && (rc = odo_timed_event(ctx, o, o->stage_events, ctx->stream)) != CFEAR_OK) return rc;
  hipLaunchKernelGGL(features_cloud_step_kernel, dim3(o->B), dim3(BLOCK_F), 0, ctx->stream, d_xyi, capacity, d_counts, OP, o->d_states, o->d_scratch_hdr);
  if (o->profile && (rc = odo_timed_event(ctx, o, o->stage_events, ctx->stream)) != CFEAR_OK) return rc;
  launch_register_step(OP, o->B, ctx->stream, o);
  if (o->profile && (rc = odo_timed_event(ctx, o, o->stage_events, ctx->stream)) != CFEAR_OK) return rc;
  o->step_no++;
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

int cfear_odometry_step_cloud_device(cfear_ctx* ctx, cfear_odometry* o, const float* d_xyi, int capacity, const int* d_counts) {
  if (!ctx || !o || !d_xyi || !d_counts || capacity <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_step_cloud: bad argument");
  if (!odo_shape_ok(ctx, o)) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_step_cloud: submap_scan_size / k_strongest / filter_type changed after odometry_create, or a parity mode (cfear_tune NN_TIE_RULE / VOXEL_ORDER) was switched under the object");
  if (capacity > o->cap_points) {
    char msg[256];
    snprintf(msg, sizeof(msg), "odometry_step_cloud: capacity %d exceeds the %d points per scan this object was created for (A * k_strongest, or cfar_max_points with "
             "filter_type CA-CFAR)", capacity, o->cap_points);
    return cfear_fail(ctx, CFEAR_ERR_INVALID, msg);
  }
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  if (!o->d_flags) {  // counts beyond `capacity` are reported like every other truncation
    CFEAR_HIP_CHECK(ctx, hipMalloc(&o->d_flags, sizeof(int) * ((size_t)o->B + 1)));
    CFEAR_HIP_CHECK(ctx, hipMemsetAsync(o->d_flags, 0, sizeof(int) * ((size_t)o->B + 1), ctx->stream));
  }
  return odo_step_clouds(ctx, o, d_xyi, capacity, d_counts, false);
}

int cfear_odometry_step_device(cfear_ctx* ctx, cfear_odometry* o, const uint8_t* d_polar) {
  if (!ctx || !o || !d_polar) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_step: bad argument");
  if (!odo_shape_ok(ctx, o))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_step: submap_scan_size / k_strongest / filter_type changed after odometry_create, or a parity mode (cfear_tune NN_TIE_RULE / VOXEL_ORDER) was switched under the object");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (o->filter == CFEAR_FILTER_CACFAR) {  // radar_driver.cpp:52-56, then the cloud route
    int rc = CFEAR_OK;
    if (o->profile && (rc = odo_timed_event(ctx, o, o->filter_events, ctx->stream)) != CFEAR_OK) return rc;
    rc = odo_launch_cfar(ctx, o, d_polar, o->B, o->d_cloud, o->d_cloud_n, o->d_cfar_rows, ctx->stream);
    if (rc != CFEAR_OK) return rc;
    if (o->profile && (rc = odo_timed_event(ctx, o, o->filter_events, ctx->stream)) != CFEAR_OK) return rc;
    return odo_step_clouds(ctx, o, o->d_cloud, o->cap_points, o->d_cloud_n, true);
  }
  const OdoParams OP = odo_params(ctx, o);
  const int buf = o->overlap ? (int)(o->step_no & 1) : 0;
  hipStream_t sf = o->overlap ? o->sf : ctx->stream;
  int rc = CFEAR_OK;
  if (o->overlap) {
    CFEAR_HIP_CHECK(ctx, hipEventRecord(o->ev_in, ctx->stream));  // the sweeps are ready at this point of the context stream
    CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(sf, o->ev_in, 0));
    if (o->filt_pending[buf])  // features(t-2) of every range has read this buffer
      for (hipEvent_t e : o->ev_free[buf]) CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(sf, e, 0));
  }
  // radar_driver.cpp:58
  if (o->profile && (rc = odo_timed_event(ctx, o, o->filter_events, sf)) != CFEAR_OK) return rc;
  rc = cfear_launch_kstrongest(ctx, d_polar, o->B, o->d_slots[buf], sf);
  if (rc != CFEAR_OK) return rc;
  if (o->profile && (rc = odo_timed_event(ctx, o, o->filter_events, sf)) != CFEAR_OK) return rc;
  if (o->overlap) {
    CFEAR_HIP_CHECK(ctx, hipEventRecord(o->ev_filt[buf], sf));
    // d_polar keeps its stream-order meaning for the caller: whatever the context stream is given after this call (the
    // caller's next write into the buffer, its release to a caching allocator) waits for the filter, the only reader
    CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, o->ev_filt[buf], 0));
  }
  const int nsub = o->overlap ? o->overlap : 1;
  const int per = (o->B + nsub - 1) / nsub;
  for (int i = 0; i < nsub; i++) {
    OdoParams P = OP;
    P.seq0 = i * per;
    const int count = std::min(per, o->B - P.seq0);
    if (count <= 0) break;
    hipStream_t so = o->overlap ? o->so[i] : ctx->stream;
    if (o->overlap) CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(so, o->ev_filt[buf], 0));
    if (o->profile && (rc = odo_timed_event(ctx, o, o->stage_events, so)) != CFEAR_OK) return rc;
    if (P.phase_times)
      hipLaunchKernelGGL(features_step_kernel<true>, dim3(count), dim3(BLOCK_F), 0, so, o->d_slots[buf], ctx->d_trig, P, o->d_states,
                         o->d_scan_ptrs, o->d_scratch_hdr);
    else
      hipLaunchKernelGGL(features_step_kernel<false>, dim3(count), dim3(BLOCK_F), 0, so, o->d_slots[buf], ctx->d_trig, P, o->d_states,
                         o->d_scan_ptrs, o->d_scratch_hdr);
    if (o->overlap) CFEAR_HIP_CHECK(ctx, hipEventRecord(o->ev_free[buf][i], so));
    if (o->profile && (rc = odo_timed_event(ctx, o, o->stage_events, so)) != CFEAR_OK) return rc;
    launch_register_step(P, count, so, o);
    if (o->profile && (rc = odo_timed_event(ctx, o, o->stage_events, so)) != CFEAR_OK) return rc;
  }
  if (o->overlap) o->filt_pending[buf] = true;
  o->step_no++;
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}

int cfear_odometry_phase_times(cfear_ctx* ctx, cfear_odometry* o, int enable, long long* host_ticks /*[B][32]*/) {
  if (!ctx || !o) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_phase_times: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  const size_t bytes = sizeof(long long) * 32 * (size_t)o->B;
  if (enable >= 1 && enable <= 4) {  // (any other non-zero value: read without changing the mode)
    o->phase_detail = enable == 1 ? 1 : (enable == 4 ? 2 : 0);
    o->wg_only = enable == 3;
  }
  if (!enable) {
    if (o->d_phase_times) (void)hipFree(o->d_phase_times);
    o->d_phase_times = nullptr;
    return CFEAR_OK;
  }
  if (!o->d_phase_times) {
    if (hipMalloc(&o->d_phase_times, bytes) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc phase times");
    CFEAR_HIP_CHECK(ctx, hipMemset(o->d_phase_times, 0, bytes));
    return CFEAR_OK;
  }
  if (host_ticks) {
    CFEAR_HIP_CHECK(ctx, hipMemcpy(host_ticks, o->d_phase_times, bytes, hipMemcpyDeviceToHost));
    CFEAR_HIP_CHECK(ctx, hipMemset(o->d_phase_times, 0, bytes));
  }
  return CFEAR_OK;
}

int cfear_odometry_profile(cfear_ctx* ctx, cfear_odometry* o, int enable) {
  if (!ctx || !o) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_profile: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  o->filter_events.clear(); o->stage_events.clear(); o->pool_used = 0;  // the events go back to the pool
  if (enable && o->pool.empty()) {  // created here, outside any timed region
    for (int i = 0; i < 1024; i++) {
      hipEvent_t e = nullptr;
      CFEAR_HIP_CHECK(ctx, hipEventCreate(&e));
      o->pool.push_back(e);
    }
  }
  o->profile = enable != 0;
  return CFEAR_OK;
}

int cfear_odometry_profile_read(cfear_ctx* ctx, cfear_odometry* o, double* filter_seconds, int* filter_launches) {
  if (!ctx || !o) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_profile_read: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  double tf = 0;
  const int nf = (int)(o->filter_events.size() / 2);
  for (int i = 0; i < nf; i++) {
    float a = 0.f;
    CFEAR_HIP_CHECK(ctx, hipEventElapsedTime(&a, o->filter_events[2 * i], o->filter_events[2 * i + 1]));
    tf += a * 1e-3;
  }
  if (filter_seconds) *filter_seconds = tf;
  if (filter_launches) *filter_launches = nf;
  return CFEAR_OK;
}

int cfear_odometry_profile_read_stages(cfear_ctx* ctx, cfear_odometry* o, double* features_seconds, double* registration_seconds, int* launches) {
  if (!ctx || !o) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_profile_read_stages: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  double tf = 0, tr = 0;
  const int n = (int)(o->stage_events.size() / 3);
  for (int i = 0; i < n; i++) {
    float a = 0.f, b = 0.f;
    CFEAR_HIP_CHECK(ctx, hipEventElapsedTime(&a, o->stage_events[3 * i], o->stage_events[3 * i + 1]));
    CFEAR_HIP_CHECK(ctx, hipEventElapsedTime(&b, o->stage_events[3 * i + 1], o->stage_events[3 * i + 2]));
    tf += a * 1e-3; tr += b * 1e-3;
  }
  if (features_seconds) *features_seconds = tf;
  if (registration_seconds) *registration_seconds = tr;
  if (launches) *launches = n;
  return CFEAR_OK;
}

int cfear_odometry_step_host(cfear_ctx* ctx, cfear_odometry* o, const uint8_t* h_polar) {
  if (!ctx || !o || !h_polar) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_step_host: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t bytes = (size_t)o->B * ctx->A * ctx->R;
  if (!o->d_polar && hipMalloc(&o->d_polar, bytes + 64) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc polar batch");
  // the staging buffer is reused: the copy waits for the filter of the previous sweep (its only reader), not for that
  // sweep's features / registration
  if (o->overlap && o->step_no > 0) CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, o->ev_filt[(o->step_no - 1) & 1], 0));
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(o->d_polar, h_polar, bytes, hipMemcpyHostToDevice, ctx->stream));
  // h_polar belongs to the caller again when this returns (pinned memory makes the copy truly asynchronous)
  CFEAR_HIP_CHECK(ctx, hipEventRecord(o->ev_copied, ctx->stream));
  const int rc = cfear_odometry_step_device(ctx, o, o->d_polar);
  CFEAR_HIP_CHECK(ctx, hipEventSynchronize(o->ev_copied));
  return rc;
}

// ---- replay of a recording without a host round trip per sweep (offline_odometry.cpp:103-125) -----------------------
int cfear_host_alloc(cfear_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out || bytes == 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "host_alloc: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  *out = nullptr;
  if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipHostMalloc");
  return CFEAR_OK;
}
void cfear_host_free(cfear_ctx* ctx, void* p) {
  if (!p) return;
  if (ctx) (void)hipSetDevice(ctx->device);
  (void)hipHostFree(p);
}

// chunk: sweeps per chunk the slot buffers must hold; staging: the host route also needs the two polar staging buffers of that many
// sweeps (the device route filters the frames where they lie and never touches them: for thousands of sequences they would be
// gigabytes allocated for nothing)
static int replay_ensure(cfear_ctx* ctx, cfear_odometry* o, int chunk, bool staging, size_t n_records) {
  if (!o->rp_ready) {  // events first, the stream last, the flag only when everything exists: a failure leaves nothing half-made behind
    for (hipEvent_t* e : {&o->rp_filt[0], &o->rp_filt[1], &o->rp_used[0], &o->rp_used[1], &o->rp_in})
      if (!*e) CFEAR_HIP_CHECK(ctx, hipEventCreateWithFlags(e, hipEventDisableTiming));
    if (!o->rp_stream) {
      CFEAR_HIP_CHECK(ctx, hipStreamCreateWithFlags(&o->rp_stream, hipStreamNonBlocking));
      ctx->aux_streams.push_back(o->rp_stream);
    }
    o->rp_ready = true;
  }
  const size_t sweep = (size_t)o->B * ctx->A * ctx->R, slots = (size_t)o->B * o->cap_points;
  if (chunk > o->rp_chunk) {
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(o->rp_stream));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 2; i++) {
      for (void** q : {(void**)&o->rp_slots[i], (void**)&o->rp_cloud[i], (void**)&o->rp_cloud_n[i]}) { if (*q) (void)hipFree(*q); *q = nullptr; }
      o->rp_used_pending[i] = false;
    }
    if (o->rp_cfar_rows) (void)hipFree(o->rp_cfar_rows);
    o->rp_cfar_rows = nullptr;
    o->rp_chunk = 0;
    for (int i = 0; i < 2; i++) {
      if (o->filter == CFEAR_FILTER_CACFAR) {  // a cloud of cap_points points + its count per sweep and sequence
        if (hipMalloc(&o->rp_cloud[i], sizeof(float) * 3 * slots * chunk) != hipSuccess || hipMalloc(&o->rp_cloud_n[i], sizeof(int) * (size_t)o->B * chunk) != hipSuccess)
          return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc replay cloud buffers");
      } else if (hipMalloc(&o->rp_slots[i], sizeof(uint32_t) * slots * chunk) != hipSuccess) {
        return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc replay slot buffers");
      }
    }
    if (o->filter == CFEAR_FILTER_CACFAR && hipMalloc(&o->rp_cfar_rows, sizeof(int) * cfear_cfar_scratch_ints(ctx, (size_t)o->B * chunk)) != hipSuccess)
      return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc replay cfar rows");
    o->rp_chunk = chunk;
  }
  if (staging && chunk > o->rp_polar_chunk) {
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(o->rp_stream));
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 2; i++) {
      if (o->rp_polar[i]) (void)hipFree(o->rp_polar[i]);
      o->rp_polar[i] = nullptr;
    }
    o->rp_polar_chunk = 0;
    for (int i = 0; i < 2; i++)
      if (hipMalloc(&o->rp_polar[i], sweep * chunk + 64) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc replay staging");
    o->rp_polar_chunk = chunk;
  }
  if (n_records > o->records_cap) {
    CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (o->d_records) (void)hipFree(o->d_records);
    o->d_records = nullptr; o->records_cap = 0;
    if (hipMalloc(&o->d_records, sizeof(cfear_sweep_record) * n_records) != hipSuccess) return cfear_fail(ctx, CFEAR_ERR_NOMEM, "hipMalloc sweep records");
    o->records_cap = n_records;
  }
  return CFEAR_OK;
}

// frames: n_sweeps x B x A x R bytes on the host (copied chunk by chunk into the staging buffers) or on the device (filtered where
// they lie); d_records: where the per-sweep records go on the device (null: none)
static int replay_impl_queue(cfear_ctx* ctx, cfear_odometry* o, const uint8_t* frames, bool on_device, int n_sweeps, cfear_sweep_record* d_records) {
  const size_t sweep = (size_t)o->B * ctx->A * ctx->R, slots = (size_t)o->B * o->cap_points;
  // chunk: enough sweeps for the filter to run at its streaming rate (>= ~16 k azimuth rows per launch) and for the copy of the
  // next chunk to hide behind the odometry kernels of this one, at most 256 MB per buffer (host route: of staged sweeps; device
  // route: of filter slots - there is no staging)
  const bool cfar = o->filter == CFEAR_FILTER_CACFAR;
  const size_t filt_bytes = cfar ? sizeof(float) * 3 * slots : sizeof(uint32_t) * slots;  // the filter's output per sweep
  const size_t per_sweep = on_device ? filt_bytes : std::max(sweep, filt_bytes);
  int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)64, ((size_t)256 << 20) / per_sweep));
  chunk = std::min(chunk, n_sweeps);
  {  // buffers of an earlier call may hold more sweeps per chunk: at least as good
    const int have = on_device ? o->rp_chunk : std::min(o->rp_chunk, o->rp_polar_chunk);
    chunk = std::max(chunk, std::min(have, n_sweeps));
  }
  int rc = replay_ensure(ctx, o, chunk, !on_device, 0);
  if (rc != CFEAR_OK) return rc;
  const int nchunks = (n_sweeps + chunk - 1) / chunk;
  OdoParams OP = odo_params(ctx, o);
  const bool persistent = o->B <= ctx->tune_replay_persistent_max && !OP.phase_times && !OP.wg_times;
  if (on_device) {  // the sweeps are ready at this point of the context stream: the replay stream (which reads them) starts there
    CFEAR_HIP_CHECK(ctx, hipEventRecord(o->rp_in, ctx->stream));
    CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(o->rp_stream, o->rp_in, 0));
  }
  auto stage = [&](int c) -> int {  // copy + filter of chunk c on the replay stream
    const int b = c & 1, t0 = c * chunk, cnt = std::min(chunk, n_sweeps - t0);
    if (o->rp_used_pending[b]) CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(o->rp_stream, o->rp_used[b], 0));  // its slots were consumed
    const uint8_t* src = frames + sweep * (size_t)t0;
    if (!on_device) {
      CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(o->rp_polar[b], src, sweep * (size_t)cnt, hipMemcpyHostToDevice, o->rp_stream));
      src = o->rp_polar[b];
    }
    const int frc = cfar ? odo_launch_cfar(ctx, o, src, cnt * o->B, o->rp_cloud[b], o->rp_cloud_n[b], o->rp_cfar_rows, o->rp_stream)  // radar_driver.cpp:52-56
                         : cfear_launch_kstrongest(ctx, src, cnt * o->B, o->rp_slots[b], o->rp_stream);  // radar_driver.cpp:58, pose-independent
    if (frc != CFEAR_OK) return frc;
    CFEAR_HIP_CHECK(ctx, hipEventRecord(o->rp_filt[b], o->rp_stream));
    return CFEAR_OK;
  };
  rc = stage(0);
  if (rc != CFEAR_OK) return rc;
  for (int c = 0; c < nchunks; c++) {
    const int b = c & 1, t0 = c * chunk, cnt = std::min(chunk, n_sweeps - t0);
    CFEAR_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, o->rp_filt[b], 0));
    // odometrykeyframefuser.cpp:143-259 sweep after sweep. Few sequences: one persistent workgroup per sequence walks the whole
    // chunk (no launch in between; a workgroup needs a compute unit's LDS to itself, so this pays while the sequences fit the
    // chip at one per compute unit). Many sequences: the batched kernels, two launches per sweep, whose occupancy is what counts.
    if (persistent && cfar) {
      cfear_launch_replay_chunk_cloud(o->rp_cloud[b], o->cap_points, o->rp_cloud_n[b], cnt, o->B, &OP, o->d_states, o->d_scratch_hdr, o->d_cov_work, o->d_summaries,
                                      o->d_poses_out, d_records ? d_records + (size_t)t0 * o->B : nullptr, ctx->stream);
    } else if (persistent) {
      cfear_launch_replay_chunk(o->rp_slots[b], cnt, o->B, ctx->d_trig, &OP, o->d_states, o->d_scratch_hdr, o->d_cov_work, o->d_summaries,
                                o->d_poses_out, d_records ? d_records + (size_t)t0 * o->B : nullptr, ctx->stream);
    } else {
      for (int t = 0; t < cnt; t++) {
        OP.records = d_records ? d_records + (size_t)(t0 + t) * o->B : nullptr;
        if (cfar) odo_launch_sweep_cloud(o, OP, o->rp_cloud[b] + 3 * slots * (size_t)t, o->cap_points, o->rp_cloud_n[b] + (size_t)o->B * t, o->B, ctx->stream);
        else odo_launch_sweep(ctx, o, OP, o->rp_slots[b] + slots * (size_t)t, o->B, ctx->stream);
      }
    }
    CFEAR_HIP_CHECK(ctx, hipEventRecord(o->rp_used[b], ctx->stream));
    o->rp_used_pending[b] = true;
    o->step_no += cnt;
    if (c + 1 < nchunks && (rc = stage(c + 1)) != CFEAR_OK) return rc;  // (queued after this chunk's launches: a copy from pageable memory blocks the host)
  }
  CFEAR_HIP_CHECK(ctx, hipGetLastError());
  return CFEAR_OK;
}
static int replay_impl(cfear_ctx* ctx, cfear_odometry* o, const uint8_t* frames, bool on_device, int n_sweeps, cfear_sweep_record* d_records) {
  const int rc = replay_impl_queue(ctx, o, frames, on_device, n_sweeps, d_records);
  if (rc != CFEAR_OK && o->rp_stream) {
    // an error after work was queued: copies out of the caller's frames and kernels that read them may be in flight - nothing of
    // this call is left running when it returns (the error text is kept)
    const std::string keep = ctx->err;
    (void)hipStreamSynchronize(o->rp_stream);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->err = keep;
  }
  return rc;
}

static int replay_check(cfear_ctx* ctx, cfear_odometry* o, const uint8_t* frames, int n_sweeps) {
  if (!ctx || !o || !frames || n_sweeps <= 0) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_replay: bad argument");
  if (!odo_shape_ok(ctx, o))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_replay: submap_scan_size / k_strongest / filter_type changed after odometry_create, or a parity mode (cfear_tune NN_TIE_RULE / VOXEL_ORDER) was switched under the object");
  // CA-CFAR reads the images as dwords: every chunk of the replay starts a whole number of sweeps (of B images) after `frames`, so the base and the
  // sweep size decide the alignment of all of them - refused here, before any chunk has advanced the sequences' state
  if (o->filter == CFEAR_FILTER_CACFAR && ((reinterpret_cast<uintptr_t>(frames) & 3) != 0 || (n_sweeps > 1 && (((size_t)o->B * ctx->A * ctx->R) & 3) != 0)))
    return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_replay: filter_type CA-CFAR needs a 4-byte aligned recording and B * A * R a multiple of 4 (the detector reads whole dwords)");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return odo_join(ctx, o);
}

int cfear_odometry_replay_host(cfear_ctx* ctx, cfear_odometry* o, const uint8_t* h_frames, int n_sweeps, cfear_sweep_record* records) {
  int rc = replay_check(ctx, o, h_frames, n_sweeps);
  if (rc != CFEAR_OK) return rc;
  if (records && (rc = replay_ensure(ctx, o, 0, false, (size_t)n_sweeps * o->B)) != CFEAR_OK) return rc;
  rc = replay_impl(ctx, o, h_frames, false, n_sweeps, records ? o->d_records : nullptr);
  if (rc != CFEAR_OK) return rc;
  if (records) CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(records, o->d_records, sizeof(cfear_sweep_record) * (size_t)n_sweeps * o->B, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return odo_capacity_check(ctx, o, "odometry_replay_host");
}

int cfear_odometry_replay_device(cfear_ctx* ctx, cfear_odometry* o, const uint8_t* d_frames, int n_sweeps, cfear_sweep_record* d_records) {
  const int rc = replay_check(ctx, o, d_frames, n_sweeps);
  if (rc != CFEAR_OK) return rc;
  return replay_impl(ctx, o, d_frames, true, n_sweeps, d_records);  // asynchronous: nothing is waited for
}

// after a synchronisation of the context stream: has any scan of this object overflowed its cell capacity (CFEAR_TUNE_MAX_CELLS)?
static int odo_capacity_check(cfear_ctx* ctx, cfear_odometry* o, const char* what) {
  if (!o->d_flags) return CFEAR_OK;  // sized for every filtered point: cannot happen
  int f = 0;
  CFEAR_HIP_CHECK(ctx, hipMemcpy(&f, o->d_flags, sizeof(int), hipMemcpyDeviceToHost));
  if (f & 2) {
    char msg[320];
    snprintf(msg, sizeof(msg), "%s: a sweep's cloud had more points than the %d this object is sized for (cfear_params.cfar_max_points with filter_type CA-CFAR; the "
             "capacity argument of cfear_odometry_step_cloud_device): the first %d in (azimuth, range) order were kept, the results of that sequence are those "
             "of a truncated cloud", what, o->cap_points, o->cap_points);
    return cfear_fail(ctx, CFEAR_ERR_CAPACITY, msg);
  }
  if (f & 1) {
    char msg[400];
    snprintf(msg, sizeof(msg), "%s: a scan produced more than %d oriented surface points (%s): its first %d were kept, the results of that sequence are those of a "
             "truncated scan", what, o->cap_cells, ctx->tune_max_cells > 0 ? "cfear_tune CFEAR_TUNE_MAX_CELLS of this object" :
             "the default of objects with more than 7 keyframes: min(points per scan, 4096); raise it with cfear_tune CFEAR_TUNE_MAX_CELLS before cfear_odometry_create", o->cap_cells);
    return cfear_fail(ctx, CFEAR_ERR_CAPACITY, msg);
  }
  return CFEAR_OK;
}

int cfear_odometry_poses(cfear_ctx* ctx, cfear_odometry* o, double* poses_xyt) {
  if (!ctx || !o || !poses_xyt) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_poses: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(poses_xyt, o->d_poses_out, sizeof(double) * 3 * (size_t)o->B, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return odo_capacity_check(ctx, o, "odometry_poses");
}

int cfear_odometry_covariances(cfear_ctx* ctx, cfear_odometry* o, double* cov6) {
  if (!ctx || !o || !cov6) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_covariances: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  CFEAR_HIP_CHECK(ctx, hipMemcpyAsync(cov6, o->d_cov_work, sizeof(double) * 36 * (size_t)o->B, hipMemcpyDeviceToHost, ctx->stream));
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return odo_capacity_check(ctx, o, "odometry_covariances");
}

int cfear_odometry_status(cfear_ctx* ctx, cfear_odometry* o, int32_t* per_sequence) {
  if (!ctx || !o) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_status: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (per_sequence) {
    if (o->d_flags) CFEAR_HIP_CHECK(ctx, hipMemcpy(per_sequence, o->d_flags + 1, sizeof(int) * (size_t)o->B, hipMemcpyDeviceToHost));
    else memset(per_sequence, 0, sizeof(int) * (size_t)o->B);
  }
  return odo_capacity_check(ctx, o, "odometry_status");
}

int cfear_odometry_summary(cfear_ctx* ctx, cfear_odometry* o, int sequence, cfear_reg_summary* summary, int* n_cells,
                           int* n_keyframes) {
  if (!ctx || !o || sequence < 0 || sequence >= o->B) return cfear_fail(ctx, CFEAR_ERR_INVALID, "odometry_summary: bad argument");
  CFEAR_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  { const int jrc = odo_join(ctx, o); if (jrc != CFEAR_OK) return jrc; }
  CFEAR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (summary) CFEAR_HIP_CHECK(ctx, hipMemcpy(summary, o->d_summaries + sequence, sizeof(cfear_reg_summary), hipMemcpyDeviceToHost));
  if (n_cells || n_keyframes) {
    SeqState st;
    CFEAR_HIP_CHECK(ctx, hipMemcpy(&st, o->d_states + sequence, sizeof(st), hipMemcpyDeviceToHost));
    if (n_keyframes) *n_keyframes = st.nkf;
    if (n_cells) {  // cells of the scan built by the last step
      const ScanLayout SL = scan_layout(o->cap_points, o->cap_cells, false, o->with_kd);
      ScanDev h;
      CFEAR_HIP_CHECK(ctx, hipMemcpy(&h, o->d_scans + SL.total * ((size_t)sequence * o->nslots + st.last_slot), sizeof(h), hipMemcpyDeviceToHost));
      *n_cells = h.n_cells;
    }
  }
  return odo_capacity_check(ctx, o, "odometry_summary");
}

}  // extern "C"
