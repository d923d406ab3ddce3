// registration_dev.h -- K6/K7/K8 device code: scan-to-keyframes registration of one problem by one
// workgroup, entirely on the device (association, robust normal equations, Levenberg-Marquardt
// controller, outer re-association loop, covariance). Replaces n_scan_normal_reg::Register
// (n_scan_normal.cpp:82-187) + AddScanPairCost (:215-326) + ceres::Solve (:443-452) + GetCovariance
// (:392-433) for this path. Ceres trust-region LM semantics restated per SURVEY.md 9.H; same
// formulas, in double, as oracle/cfear_oracle.c so that iteration counts agree.
//
// Structure: the solver state lives in LDS (RegShared). Wave 0 is the controller: a state machine of leaf functions
// (ctl_*) that consumes the result of the command just executed and publishes the next one (BUILD: re-associate at the
// current pose; EVAL: robustified cost / gradient / Gauss-Newton matrix at a point); all waves execute the commands
// (association: thread <-> source cell; evaluation: thread <-> residual block), each followed by a deterministic
// reduction. Everything reaches RegShared through an LDS-typed pointer (LRegShared).
#pragma once
#include "features_dev.h"

namespace cfear_dev {

struct RegParams {
  int cost, loss, weight_opt;
  int recompute_repeats;  // 1: an outer iteration that would repeat the previous one exactly is run again anyway (ctl_lm_done; cfear_tune REPEAT_SHORTCUT = 0)
  double loss_limit, covar_scale, regularization, assoc_radius;
  int max_outer, min_itr, max_inner;
  int nn_tie;  // cfear_tune NN_TIE_RULE: 0 = exact-distance 1-NN ties to the lowest cell index (production), 1 = highest, 2 = FLANN's kd-tree order:
               // the non-zero rules take the general association path (slow: sensitivity / parity modes)
};

// Per-block global scratch: compacted matches (SoA) + per-pair association result.
struct RegScratch {
  double* tmx; double* tmy;  // Ttar * tar_mean
  double* a0; double* a1; double* a2;  // P2L: (n_x, n_y, -) ; P2D: (l00, l10, l11)
  double* sx; double* sy;    // source mean (local frame)
  double* w;                 // weight after loss
  int* assoc;                // [acap] ints: the general path's target cell index per pair; the grouped path's parked matches (four ints per
                             // (group of four keyframes, source cell)) and, behind them, their positions (two ints each)
  int cap;                   // capacity of the match arrays in pairs
  int acap;                  // ints of `assoc`
  double* red;               // LDS, >= 10 * 32 doubles
  int* red_i;                // LDS, >= 64 ints (also used as 32 x u64 scan scratch)
};

struct Aff2 { double l0, l1, l2, l3, t0, t1; };

__device__ inline Aff2 aff_from_xyt(double x, double y, double th) {  // vectorToAffine3d, registration.cpp:130-136
  Aff2 T; double s, c; sincos(th, &s, &c);
  T.l0 = c; T.l1 = -s; T.l2 = s; T.l3 = c; T.t0 = x; T.t1 = y;
  return T;
}
__device__ inline Aff2 aff_mul(const Aff2& A, const Aff2& B) {
  Aff2 C;
  C.l0 = A.l0 * B.l0 + A.l1 * B.l2; C.l1 = A.l0 * B.l1 + A.l1 * B.l3;
  C.l2 = A.l2 * B.l0 + A.l3 * B.l2; C.l3 = A.l2 * B.l1 + A.l3 * B.l3;
  C.t0 = (A.l0 * B.t0 + A.l1 * B.t1) + A.t0;
  C.t1 = (A.l2 * B.t0 + A.l3 * B.t1) + A.t1;
  return C;
}
__device__ inline Aff2 aff_inv(const Aff2& A) {
  Aff2 I;
  const double det = A.l0 * A.l3 - A.l1 * A.l2;
  const double id = 1.0 / det;
  I.l0 = A.l3 * id; I.l1 = -A.l1 * id; I.l2 = -A.l2 * id; I.l3 = A.l0 * id;
  I.t0 = -(I.l0 * A.t0 + I.l1 * A.t1);
  I.t1 = -(I.l2 * A.t0 + I.l3 * A.t1);
  return I;
}
__device__ inline void aff_to_xyt(const Aff2& T, double v[3]) {  // Affine3dToVectorXYeZ, utils.cpp:115-122
  v[0] = T.t0; v[1] = T.t1; v[2] = atan2(T.l2, T.l3);
}
__device__ inline Aff2 aff_identity() { Aff2 T; T.l0 = 1; T.l1 = 0; T.l2 = 0; T.l3 = 1; T.t0 = 0; T.t1 = 0; return T; }

// quotient of well-scaled positive numbers (sample counts >= 6, planarity scales >= log 1.5): hardware reciprocal, two Newton
// steps, one residual correction - the quotient without the scaling / special-case handling of a full IEEE division (a third
// of its instructions; two of these per residual block); agrees with it to an ulp
__device__ __forceinline__ double div_well_scaled(double num, double den) {
  double r = __builtin_amdgcn_rcp(den);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
  const double q = num * r;
  return __builtin_fma(__builtin_fma(-den, q, num), r, q);
}
__device__ inline double similarity(double x, double y) { return div_well_scaled(2 * fmin(x, y), x + y); }  // registration.h:96
// square root of a well-scaled non-negative number (norms of steps and poses): hardware reciprocal square root, two Newton
// steps, one correction of the root; half the instructions of the IEEE sequence, within an ulp or two of it
__device__ __forceinline__ double sqrt_well_scaled(double s) {
  double y = __builtin_amdgcn_rsq(s > 0.0 ? s : 1.0);
  y = y * __builtin_fma(-0.5 * s, y * y, 1.5);
  y = y * __builtin_fma(-0.5 * s, y * y, 1.5);
  const double r = s * y;
  return s > 0.0 ? __builtin_fma(0.5 * y, __builtin_fma(-r, r, s), r) : 0.0;
}
// natural logarithm of a finite x >= 1 (the Cauchy loss takes log(1 + s / a^2)): x = m 2^e with m in [sqrt(1/2), sqrt(2)), log m =
// 2 atanh((m - 1) / (m + 1)) by its series in z^2 <= 0.0295 (ten terms: 2e-17), e ln 2 added in two parts. About forty instructions where
// the library's correctly rounded logarithm takes twice as many (and a thousand residual blocks per keyframe pair evaluate it ~45 times
// per registration); within an ulp or two of it - the same order as the summation-order differences between this code and the oracle.
__device__ __forceinline__ double log_ge1(double x) {
  int e = __builtin_amdgcn_frexp_exp(x);
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  e = lo ? e - 1 : e;
  const double f = m - 1.0;
  const double z = div_well_scaled(f, m + 1.0), z2 = z * z;
  double p = 1.0 / 21.0;
  p = __builtin_fma(p, z2, 1.0 / 19.0); p = __builtin_fma(p, z2, 1.0 / 17.0); p = __builtin_fma(p, z2, 1.0 / 15.0);
  p = __builtin_fma(p, z2, 1.0 / 13.0); p = __builtin_fma(p, z2, 1.0 / 11.0); p = __builtin_fma(p, z2, 1.0 / 9.0);
  p = __builtin_fma(p, z2, 1.0 / 7.0); p = __builtin_fma(p, z2, 1.0 / 5.0); p = __builtin_fma(p, z2, 1.0 / 3.0);
  const double lm = __builtin_fma(z * z2, p + p, z + z);  // 2 z + 2 z^3 (1/3 + ...)
  const double ef = (double)e;
  return __builtin_fma(ef, 6.93147180369123816490e-01, __builtin_fma(ef, 1.90821492927058770002e-10, lm));  // ln 2 = hi + lo
}
__device__ inline double get_weight(int opt, double n1, double n2, double sim, double p1, double p2) {  // registration.cpp:67-76
  switch (opt) {
    case 0: return 1.0;
    case 1: return similarity(n1, n2);
    case 2: return sim;
    case 3: return similarity(p1, p2);
    case 4: return similarity(n1, n2) + sim + similarity(p1, p2);
    default: return 1.0;
  }
}

#define CFEAR_DBL_MIN 2.2250738585072014e-308
// ceres::LossFunction::Evaluate restatement (registration.cpp:78-97)
struct Rho { double v, d1; };  // rho(s), rho'(s), returned in registers
__device__ __noinline__ Rho loss_eval(int loss, double a, double s) {
  const double b = a * a;
  Rho o;
  switch (loss) {
    case CFEAR_LOSS_HUBER:
      if (s > b) { const double r = sqrt(s); o.v = 2.0 * a * r - b; o.d1 = fmax(CFEAR_DBL_MIN, a / r); }
      else { o.v = s; o.d1 = 1.0; }
      return o;
    case CFEAR_LOSS_CAUCHY: {
      const double c = 1.0 / b, sum = 1.0 + s * c, inv = 1.0 / sum;
      o.v = b * log(sum); o.d1 = fmax(CFEAR_DBL_MIN, inv);
      return o; }
    case CFEAR_LOSS_SOFTLONE: {
      const double c = 1.0 / b, sum = 1.0 + s * c, tmp = sqrt(sum);
      o.v = 2.0 * b * (tmp - 1.0); o.d1 = fmax(CFEAR_DBL_MIN, 1.0 / tmp);
      return o; }
    case CFEAR_LOSS_TUKEY:
      if (s <= b) { const double v = 1.0 - s / b, v2 = v * v; o.v = b / 3.0 * (1.0 - v2 * v); o.d1 = v2; }
      else { o.v = b / 3.0; o.d1 = 0.0; }
      return o;
    case CFEAR_LOSS_COMBINED: {
      double g0, g1, f0, f1;
      { const double sum = 1.0 + s, inv = 1.0 / sum; g0 = log(sum); g1 = fmax(CFEAR_DBL_MIN, inv); }
      if (g0 > 1.0) { const double r = sqrt(g0); f0 = 2.0 * r - 1.0; f1 = fmax(CFEAR_DBL_MIN, 1.0 / r); }
      else { f0 = g0; f1 = 1.0; }
      o.v = f0; o.d1 = f1 * g1;
      return o; }
    default: o.v = s; o.d1 = 1.0; return o;
  }
}

struct NormalEq { double cost, g0, g1, g2, h00, h01, h02, h11, h12, h22; };

struct SolveSummary { int num_iterations; int termination; double final_cost; double last_relative_decrease; };

#ifndef CFEAR_REG_MAX_SCANS
#define CFEAR_REG_MAX_SCANS 64  // scans (keyframes + current) a registration of this translation unit can have: sizes the per-scan arrays
                                // of RegShared. register_step.hip compiles the batched step kernel for 8 (submap_scan_size <= 7: every
                                // preset of the reference) and spends the 10 KB of LDS that frees on the match array
#endif
#define CFEAR_RED_STRIDE 8  // partial sums of up to 8 waves per quantity (W.red)
#ifndef CFEAR_EVAL_WAVES
#define CFEAR_EVAL_WAVES 4  // waves that evaluate residuals (one per SIMD); the rest only keep the barriers (register_step_large.hip: all eight of its
                            // 512-thread workgroups - a fifty-keyframe submap evaluates ~6000 residual blocks ~45 times per registration)
#endif
#ifndef CFEAR_REG_BLOCK
#define CFEAR_REG_BLOCK 256 // threads of every workgroup that runs this code (pipeline.hip BLOCK_R; replay.hip compiles it for 512): a compile-time constant,
                            // because blockDim.x is a load from the dispatch packet - a round trip to memory wherever an
                            // out-of-line function asks for it (the evaluation did, thirteen times per registration)
#endif

enum { REG_CMD_BUILD = 1, REG_CMD_EVAL = 2, REG_CMD_DONE = 3 };
enum { REG_ST_BUILD = 0, REG_ST_LM_IT0 = 1, REG_ST_LM_CAND = 2, REG_ST_COV = 3 };

struct RegIo {  // where the controller reads/writes the caller-visible data
  double* poses; double* cov6; cfear_reg_summary* out; double* par; int n;
};

// Block-shared state of one registration (LDS). The controller fields are only touched by wave 0.
struct RegShared {
  // command published by the controller (wave 0) to all waves
  int cmd, itr, M, state;
  int lds_match, assoc_path;  // assoc_path: which association path the last build took (cfear_reg_summary::reserved: 1 one block of cells against <= 4 keyframes,
                              // 2 (group, cell) items dealt densely, 3 pair ranges - the general path); lds_match: where the compacted matches live: 1 the LDS match array (M <= match_lds_cap(cost)), 2 its capacity there and the rest in memory, 0 memory
  double x[3];  // parameters to evaluate at (EVAL) / current pose of the last scan (BUILD)
  double c, s;  // cos/sin of x[2], computed once by the controller
  double cur_c, cur_s, prev_c, prev_s;  // cos/sin of xcur[2] and prev_par[2]: the values published with the evaluation that produced
                                        // them, so that a re-association / first evaluation at xcur needs no sincos of its own
  double Ttar[CFEAR_REG_MAX_SCANS][6];  // keyframe poses as affine maps (vectorToAffine3d, registration.cpp:130-136)
  double Trel[CFEAR_REG_MAX_SCANS][6];  // Ttar^-1 * Tsrc (n_scan_normal.cpp:224)
  GridView kf[CFEAR_REG_MAX_SCANS];      // 1-NN search view of every scan (filled once per Register call)
  const double* srs; long long scc;      // source scan (the last one): its SoA cell view (ScanDev::rsrc) and that array's stride
  // parameter blocks the out-of-line functions take by reference: kept here so that they are LDS reads, not reads
  // of a per-thread stack copy
  RegParams rp; RegScratch rw; RegIo rio;
  // soft constraint (Register(..., soft_constraints = true), n_scan_normal.cpp:373-377): residual L alpha (guess - x)
  int prior_on, pad_p; double pL[9], pguess[3], palpha;
  // ---- controller state: outer association loop (n_scan_normal.cpp:82-187)
  int success, nres, ret, pad0;
  double xcur[3], prev_par[3], tsrc_last[3], prev_score;
  SolveSummary ss;
  // ---- controller state: Levenberg-Marquardt (ceres::Solve restatement)
  NormalEq E;
  NormalEq G;  // sums of the evaluation just done (gather_partials): an 80-byte struct returned by value from an out-of-line
               // function travels through per-thread scratch, a round trip to memory on the controller's serial chain
  double x_cost, x_norm, sc0, sc1, sc2, radius, decrease_factor, dg0, dg1, dg2, xc[3], model_cost_change;
  int reuse_diagonal, num_invalid, iteration, nrec;  // nrec: outer iterations recorded in orec
  int moved, pad_m;  // the current solve has accepted a step (the pose differs from the one its problem was built at)
  // per-outer-iteration summary (cfear_reg_summary::inner_iterations ...) of the first CFEAR_OUTER_LDS iterations, written to
  // memory once at the end: a store to memory in an out-of-line controller function is a memory round trip on the serial
  // chain (the calling convention waits for it at the return) - 1.8 us per outer iteration
  struct OuterRec { int inner, term; double cost, pose[3]; } orec[8];
};
#define CFEAR_OUTER_LDS 8

// RegShared lives in LDS. Through a generic pointer every field access converts the address (64-bit add, compare with
// null, select: three extra instructions each, a third of the controller's instruction stream); the functions below take
// it through an LDS-typed pointer instead. Structs are copied field by field (no copy constructors across address spaces).
typedef __attribute__((address_space(3))) RegShared LRegShared;
typedef __attribute__((address_space(3))) NormalEq LNormalEq;
typedef __attribute__((address_space(3))) double lds_f64;
__device__ __forceinline__ NormalEq neq_load(const LNormalEq* p) {
  NormalEq e; e.cost = p->cost; e.g0 = p->g0; e.g1 = p->g1; e.g2 = p->g2; e.h00 = p->h00; e.h01 = p->h01; e.h02 = p->h02;
  e.h11 = p->h11; e.h12 = p->h12; e.h22 = p->h22;
  return e;
}
__device__ __forceinline__ void neq_store(LNormalEq* p, const NormalEq& e) {
  p->cost = e.cost; p->g0 = e.g0; p->g1 = e.g1; p->g2 = e.g2; p->h00 = e.h00; p->h01 = e.h01; p->h02 = e.h02;
  p->h11 = e.h11; p->h12 = e.h12; p->h22 = e.h22;
}
#define CFEAR_GENERIC(T, lvalue) (*(T*)&(lvalue))  // generic-address-space view of an LDS object (for by-reference parameters)

// ---- compacted matches: SoA of 8 doubles per residual block, in LDS when they fit -------------------
// 622 matches x 64 B + the rest of the registration kernels' LDS <= 53,760 B: three workgroups per compute unit need
// <= 53,760 B each (LDS is handed out in 1,280-byte granules; 53,824 B already drops the kernel to two per CU and +34 % time)
#ifndef CFEAR_MATCH_LDS_CAP
#define CFEAR_MATCH_LDS_CAP 622
#endif
#define CFEAR_MATCH_LDS_DOUBLES (8 * CFEAR_MATCH_LDS_CAP)
struct MatchPtrs { double *tmx, *tmy, *a0, *a1, *a2, *sx, *sy, *w; };
__device__ __forceinline__ MatchPtrs match_ptrs(double* base, size_t cap) {
  MatchPtrs m;
  m.tmx = base; m.tmy = base + cap; m.a0 = base + 2 * cap; m.a1 = base + 3 * cap; m.a2 = base + 4 * cap;
  m.sx = base + 5 * cap; m.sy = base + 6 * cap; m.w = base + 7 * cap;
  return m;
}
__device__ __forceinline__ double* lds_match_base() {  // one 40 KB block-shared array for every user of this header
  __shared__ double s_match[CFEAR_MATCH_LDS_DOUBLES];
  return s_match;
}
// The LDS array holds only the quantities the cost metric reads, so that more residual blocks fit: eight arrays for P2D
// (622 matches), seven for P2L (a2 is never read: 710) and five for P2P (no a0 .. a2: 995). In the synthetic yard 17 % of the
// registrations build 623-682 blocks: with eight arrays they went to the copy in memory (an evaluation of 3.5-5 us instead
// of 1.5). q = the canonical array number of match_ptrs() (tmx tmy a0 a1 a2 sx sy w); -1: not kept in LDS for this cost.
__device__ __forceinline__ constexpr int match_lds_arrays(int cost) { return cost == CFEAR_COST_P2D ? 8 : (cost == CFEAR_COST_P2L ? 7 : 5); }
// (a choice between three constants: written as DOUBLES / arrays(cost) it is an integer division at run time wherever the cost is
// not a template parameter - the emission paid for one per stored value, a quarter of its instructions)
__device__ __forceinline__ constexpr int match_lds_cap(int cost) {
  return cost == CFEAR_COST_P2D ? CFEAR_MATCH_LDS_DOUBLES / 8 : (cost == CFEAR_COST_P2L ? CFEAR_MATCH_LDS_DOUBLES / 7 : CFEAR_MATCH_LDS_DOUBLES / 5);
}
__device__ __forceinline__ constexpr int match_lds_idx(int cost, int q) {
  return cost == CFEAR_COST_P2D ? q
       : cost == CFEAR_COST_P2L ? (q < 4 ? q : (q == 4 ? -1 : q - 1))
                                : (q < 2 ? q : (q < 5 ? -1 : q - 3));
}
// (emit_cell spells these positions out)
static_assert(match_lds_idx(CFEAR_COST_P2L, 2) == 2 && match_lds_idx(CFEAR_COST_P2L, 3) == 3 && match_lds_idx(CFEAR_COST_P2L, 4) == -1 &&
              match_lds_idx(CFEAR_COST_P2L, 5) == 4 && match_lds_idx(CFEAR_COST_P2L, 6) == 5 && match_lds_idx(CFEAR_COST_P2L, 7) == 6, "P2L layout");
static_assert(match_lds_idx(CFEAR_COST_P2P, 2) == -1 && match_lds_idx(CFEAR_COST_P2P, 4) == -1 && match_lds_idx(CFEAR_COST_P2P, 5) == 2 &&
              match_lds_idx(CFEAR_COST_P2P, 6) == 3 && match_lds_idx(CFEAR_COST_P2P, 7) == 4, "P2P layout");
static_assert(match_lds_idx(CFEAR_COST_P2D, 4) == 4 && match_lds_idx(CFEAR_COST_P2D, 7) == 7, "P2D layout");
__device__ __forceinline__ MatchPtrs match_ptrs_lds(int cost) {  // null where the cost does not keep the array
  double* base = lds_match_base();
  const size_t cap = (size_t)match_lds_cap(cost);
  auto at = [&](int q) -> double* { const int i = match_lds_idx(cost, q); return i < 0 ? nullptr : base + (size_t)i * cap; };
  MatchPtrs m;
  m.tmx = at(0); m.tmy = at(1); m.a0 = at(2); m.a1 = at(3); m.a2 = at(4); m.sx = at(5); m.sy = at(6); m.w = at(7);
  return m;
}

// Robustified cost, gradient and Gauss-Newton matrix over the compacted matches at x = (x0, x1, theta)
// with (c, s) = (cos, sin)(theta). Residuals: n_scan_normal.h:190-201 (P2L), :224-243 (P2D), :336-350 (P2P);
// corrector = sqrt(rho'). Only the first CFEAR_EVAL_WAVES waves work; lane 0 of each leaves its partial
// sums in W.red[i * 32 + wave].
// SRC: where the matches are - 0: the arrays in memory, 1: the LDS array, 2: the first match_lds_cap(COST) of them in the LDS array
// and the rest in memory (a problem with more residual blocks than the LDS array holds: dense scenes)
// LOSSK: how the loss is evaluated - 1 Huber and 2 Cauchy inline (every preset of the reference uses one of the two), 0 any loss through
// loss_eval
enum { CFEAR_LOSSK_ANY = 0, CFEAR_LOSSK_HUBER = 1, CFEAR_LOSSK_CAUCHY = 2 };
template <int SRC, int COST, int LOSSK>
__device__ __forceinline__ void evaluate_partial_t(const LRegShared* ls, int M, double x0, double x1, double c, double s,
                                                   double* res_out, int res_cap) {
  const int wave = threadIdx.x >> 6;
  if (wave >= CFEAR_EVAL_WAVES) return;
  // the LDS variant reads through an LDS-typed pointer (ds_read); a generic pointer costs flat loads
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  struct Rd {
    lds_cdouble* l; const double* g; size_t cap;
    __device__ __forceinline__ double operator()(int arr, int i) const {
      if (SRC == 1) return l[match_lds_idx(COST, arr) * match_lds_cap(COST) + i];
      if (SRC == 0) return g[arr * cap + i];
      return i < match_lds_cap(COST) ? l[match_lds_idx(COST, arr) * match_lds_cap(COST) + i] : g[arr * cap + i];
    }
  } rd;
  rd.l = (lds_cdouble*)lds_match_base(); rd.g = ls->rw.tmx; rd.cap = (size_t)ls->rw.cap;
  const double loss_limit = ls->rp.loss_limit;  // parameters through the LDS-typed pointer: ds_read instead of a flat load to wait for
  const double cauchy_b = loss_limit * loss_limit, cauchy_c = 1.0 / cauchy_b;  // (used by the Cauchy instantiations only)
  const int nthr = min(CFEAR_REG_BLOCK, CFEAR_EVAL_WAVES * 64);
  NormalEq a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (!res_out) {
    // The solver's evaluations. A wave running one chain of dependent double-precision operations issues an instruction
    // every ~10 cycles, so (1) a thread takes its matches two at a time, as two independent chains; (2) the chain is kept
    // short: the corrector sqrt(rho') only ever enters the sums squared (J~^T r~ = rho' w J^T r, J~^T J~ = rho' w J^T J), so
    // that square root is not taken, and Huber's rho' = a / sqrt(s) is a reciprocal square root instead of a square root and a
    // division; (3) no branches: an absent second match runs with weight 0. (Ulps away from scaling residuals and Jacobians
    // by sqrt(rho' w) first, which is how GetCost's residuals below are still formed.)
    constexpr int nr = (COST == CFEAR_COST_P2L) ? 1 : 2;
    struct Head { double r[2], J[2][3], w2, cost; };  // what a match contributes, before it is multiplied out (few live registers)
    // the values of match i the cost reads (array order of match_ptrs(): tmx tmy a0 a1 a2 sx sy w), from LDS or from memory: one
    // branch per match, global_load (not flat) for the copy in memory
    struct Raw { double v[8]; };
    typedef __attribute__((address_space(1))) const double g_cdouble;
    auto load = [&](int i) -> Raw {
      Raw r;
      constexpr int lc = match_lds_cap(COST);
      auto from_lds = [&]() {
#pragma unroll
        for (int q = 0; q < 8; q++) r.v[q] = match_lds_idx(COST, q) >= 0 ? rd.l[(match_lds_idx(COST, q) >= 0 ? match_lds_idx(COST, q) : 0) * lc + i] : 0.0;
      };
      auto from_mem = [&]() {
        g_cdouble* g = (g_cdouble*)rd.g;
#pragma unroll
        for (int q = 0; q < 8; q++) r.v[q] = match_lds_idx(COST, q) >= 0 ? g[q * rd.cap + i] : 0.0;
      };
      if (SRC == 1) from_lds();
      else if (SRC == 0) from_mem();
      else { if (i < lc) from_lds(); else from_mem(); }
      return r;
    };
    auto head = [&](const Raw& raw, bool on) -> Head {
      Head h;
      auto rd = [&](int q, int) -> double { return raw.v[q]; };
      const int i = 0;
      const double sx = rd(5, i), sy = rd(6, i), tmx = rd(0, i), tmy = rd(1, i), wgt = on ? rd(7, i) : 0.0;
      const double px = (c * sx - s * sy) + x0;
      const double py = (s * sx + c * sy) + x1;
      const double dtx = -s * sx - c * sy;
      const double dty = c * sx - s * sy;
      if (COST == CFEAR_COST_P2L) {
        const double nx = rd(2, i), ny = rd(3, i);
        h.r[0] = (px - tmx) * nx + (py - tmy) * ny;
        h.J[0][0] = nx; h.J[0][1] = ny; h.J[0][2] = dtx * nx + dty * ny;
        h.r[1] = 0; h.J[1][0] = h.J[1][1] = h.J[1][2] = 0;
      } else if (COST == CFEAR_COST_P2D) {
        const double l00 = rd(2, i), l10 = rd(3, i), l11 = rd(4, i);
        const double dx = px - tmx, dy = py - tmy;
        h.r[0] = l00 * dx; h.r[1] = l10 * dx + l11 * dy;
        h.J[0][0] = l00; h.J[0][1] = 0; h.J[0][2] = l00 * dtx;
        h.J[1][0] = l10; h.J[1][1] = l11; h.J[1][2] = l10 * dtx + l11 * dty;
      } else {
        h.r[0] = tmx - px; h.r[1] = tmy - py;
        h.J[0][0] = -1; h.J[0][1] = 0; h.J[0][2] = -dtx;
        h.J[1][0] = 0; h.J[1][1] = -1; h.J[1][2] = -dty;
      }
      double sq = h.r[0] * h.r[0];
      if (nr == 2) sq += h.r[1] * h.r[1];
      double rho_v, rho_d1;  // rho'' <= 0 for every loss here: the corrector's alpha is 0
      if (LOSSK == CFEAR_LOSSK_HUBER) {
        const double la = loss_limit, lb = la * la;
        const bool out = sq > lb;
        const double rs = rsqrt(out ? sq : 1.0), rt = sq * rs;  // 1 / sqrt(s), sqrt(s)
        rho_v = out ? 2.0 * la * rt - lb : sq;
        rho_d1 = out ? fmax(CFEAR_DBL_MIN, la * rs) : 1.0;
      } else if (LOSSK == CFEAR_LOSSK_CAUCHY) {
        // rho = b log(1 + s / b), rho' = 1 / (1 + s / b) (ceres::CauchyLoss): 1 / b once per evaluation, the reciprocal by Newton steps
        // (the sum is >= 1), the logarithm by log_ge1 - a third of the instructions of the out-of-line general version
        const double sum = 1.0 + sq * cauchy_c;
        rho_v = cauchy_b * log_ge1(sum);
        rho_d1 = fmax(CFEAR_DBL_MIN, div_well_scaled(1.0, sum));
      } else {
        const Rho rho = loss_eval(ls->rp.loss, loss_limit, sq);
        rho_v = rho.v; rho_d1 = rho.d1;
      }
      h.cost = 0.5 * (rho_v * wgt);  // ScaledLoss (n_scan_normal.cpp:277)
      h.w2 = rho_d1 * wgt;
      return h;
    };
    auto add = [&](const Head& h) {
      a.cost += h.cost;
      if (COST == CFEAR_COST_P2P) {
        // J = [-1 0 -dtx; 0 -1 -dty] (n_scan_normal.h:336-350): the general loop below multiplies by those zeros and ones (the compiler
        // may not drop a product with zero) - 36 operations of which these 14 change a sum. Same operations in the same order on the
        // sums that do change, so the results are the general loop's bit for bit: x * -1 and x + (+-0) are exact.
        const double nw = -h.w2, j2a = h.w2 * h.J[0][2], j2b = h.w2 * h.J[1][2];
        a.g0 += nw * h.r[0]; a.g2 += j2a * h.r[0];
        a.h00 += h.w2; a.h02 += nw * h.J[0][2]; a.h22 += j2a * h.J[0][2];
        a.g1 += nw * h.r[1]; a.g2 += j2b * h.r[1];
        a.h11 += h.w2; a.h12 += nw * h.J[1][2]; a.h22 += j2b * h.J[1][2];
        return;
      }
#pragma unroll
      for (int k = 0; k < nr; k++) {
        const double j0 = h.w2 * h.J[k][0], j1 = h.w2 * h.J[k][1], j2 = h.w2 * h.J[k][2];
        a.g0 += j0 * h.r[k]; a.g1 += j1 * h.r[k]; a.g2 += j2 * h.r[k];
        a.h00 += j0 * h.J[k][0]; a.h01 += j0 * h.J[k][1]; a.h02 += j0 * h.J[k][2];
        a.h11 += j1 * h.J[k][1]; a.h12 += j1 * h.J[k][2]; a.h22 += j2 * h.J[k][2];
      }
    };
    if (SRC == 1) {
#pragma unroll 1
      for (int i = threadIdx.x; i < M; i += 2 * nthr) {
        const bool onb = i + nthr < M;
        const Head A = head(load(i), true), B = head(load(onb ? i + nthr : i), onb);
        add(A); add(B);
      }
    } else if (threadIdx.x < M) {
      // matches in memory: the loads of the next pair are issued before this pair is worked on (trip after trip, each waited
      // for its own round trip: 3.6-5.5 us per evaluation of a dense scene's 1100 blocks against 1.5 us out of LDS)
      int i = threadIdx.x;
      Raw ra = load(i), rb = load(i + nthr < M ? i + nthr : i);
#pragma unroll 1
      for (; i < M; i += 2 * nthr) {
        const int in = i + 2 * nthr;
        const int ia = in < M ? in : i, ib = in + nthr < M ? in + nthr : ia;
        const Raw na = load(ia), nb = load(ib);
        const bool onb = i + nthr < M;
        const Head A = head(ra, true), B = head(rb, onb);
        add(A); add(B);
        ra = na; rb = nb;
      }
    }
  } else
  for (int i = threadIdx.x; i < M; i += nthr) {  // GetCost: also the robustified residuals, in residual-block order
    const double sx = rd(5, i), sy = rd(6, i), tmx = rd(0, i), tmy = rd(1, i), wgt = rd(7, i);
    const double px = (c * sx - s * sy) + x0;
    const double py = (s * sx + c * sy) + x1;
    const double dtx = -s * sx - c * sy;
    const double dty = c * sx - s * sy;
    double r[2], J[2][3];
    int nr;
    if (COST == CFEAR_COST_P2L) {
      const double nx = rd(2, i), ny = rd(3, i);
      nr = 1;
      r[0] = (px - tmx) * nx + (py - tmy) * ny;
      J[0][0] = nx; J[0][1] = ny; J[0][2] = dtx * nx + dty * ny;
      r[1] = 0; J[1][0] = J[1][1] = J[1][2] = 0;
    } else if (COST == CFEAR_COST_P2D) {
      const double l00 = rd(2, i), l10 = rd(3, i), l11 = rd(4, i);
      nr = 2;
      const double dx = px - tmx, dy = py - tmy;
      r[0] = l00 * dx; r[1] = l10 * dx + l11 * dy;
      J[0][0] = l00; J[0][1] = 0; J[0][2] = l00 * dtx;
      J[1][0] = l10; J[1][1] = l11; J[1][2] = l10 * dtx + l11 * dty;
    } else {
      nr = 2;
      r[0] = tmx - px; r[1] = tmy - py;
      J[0][0] = -1; J[0][1] = 0; J[0][2] = -dtx;
      J[1][0] = 0; J[1][1] = -1; J[1][2] = -dty;
    }
    double sq = r[0] * r[0];
    if (nr == 2) sq += r[1] * r[1];
    Rho rho;
    if (LOSSK == CFEAR_LOSSK_HUBER) {
      const double la = loss_limit, lb = la * la;
      if (sq > lb) { const double r = sqrt(sq); rho.v = 2.0 * la * r - lb; rho.d1 = fmax(CFEAR_DBL_MIN, la / r); }
      else { rho.v = sq; rho.d1 = 1.0; }
    } else {
      rho = loss_eval(ls->rp.loss, loss_limit, sq);
    }
    a.cost += 0.5 * (rho.v * wgt);  // ScaledLoss (n_scan_normal.cpp:277)
    const double sr = sqrt(rho.d1 * wgt);
    for (int k = 0; k < nr; k++)
      if (i * nr + k < res_cap) res_out[i * nr + k] = sr * r[k];
    for (int k = 0; k < nr; k++) {
      const double rk = sr * r[k];
      const double j0 = sr * J[k][0], j1 = sr * J[k][1], j2 = sr * J[k][2];
      a.g0 += j0 * rk; a.g1 += j1 * rk; a.g2 += j2 * rk;
      a.h00 += j0 * j0; a.h01 += j0 * j1; a.h02 += j0 * j2;
      a.h11 += j1 * j1; a.h12 += j1 * j2; a.h22 += j2 * j2;
    }
  }
  const double v[10] = {a.cost, a.g0, a.g1, a.g2, a.h00, a.h01, a.h02, a.h11, a.h12, a.h22};
  double t[3];
  wave_sum10_transposed(v, t);  // lanes 60..63 end up with the ten wave totals between them (blockops.h)
  const int lane = lane_id();
  if (lane >= 60) {
    auto* red = CFEAR_LDS_PTR(double, ls->rw.red);  // ds_write, not flat stores
    const int first = lane == 60 ? 0 : (lane == 62 ? 3 : (lane == 61 ? 5 : 8));  // index of the lane's first sum
    red[first * CFEAR_RED_STRIDE + wave] = t[0];
    red[(first + 1) * CFEAR_RED_STRIDE + wave] = t[1];
    if ((lane & 2) == 0) red[(first + 2) * CFEAR_RED_STRIDE + wave] = t[2];
  }
}
template <int COST, int LOSSK>
__device__ __noinline__ void evaluate_partial_c(const LRegShared* ls, int M, int lds_match, double x0, double x1, double c, double s,
                                                double* res_out, int res_cap) {
  if (lds_match == 1) evaluate_partial_t<1, COST, LOSSK>(ls, M, x0, x1, c, s, res_out, res_cap);
  else if (lds_match == 2) evaluate_partial_t<2, COST, LOSSK>(ls, M, x0, x1, c, s, res_out, res_cap);
  else evaluate_partial_t<0, COST, LOSSK>(ls, M, x0, x1, c, s, res_out, res_cap);
}
// KCOST: the cost metric when the kernel is compiled for one (CFEAR_COST_*; the batched registration kernel exists once per cost:
// no dispatch, and Huber with the matches in LDS - every preset of the reference - evaluates inline whatever the cost), -1: read
// from the parameters at run time (per-call API, the replay kernel)
template <int KCOST = -1>
__device__ __forceinline__ void evaluate_partial(const LRegShared* ls, int M, int lds_match, double x0, double x1, double c, double s,
                                                 double* res_out = nullptr, int res_cap = 0) {
  const int loss = ls->rp.loss;
  if (KCOST >= 0) {
    constexpr int KC = KCOST >= 0 ? KCOST : CFEAR_COST_P2L;
    if (loss == CFEAR_LOSS_HUBER) {
      if (lds_match == 1 && !res_out) evaluate_partial_t<1, KC, CFEAR_LOSSK_HUBER>(ls, M, x0, x1, c, s, nullptr, 0);
      else evaluate_partial_c<KC, CFEAR_LOSSK_HUBER>(ls, M, lds_match, x0, x1, c, s, res_out, res_cap);
    } else if (loss == CFEAR_LOSS_CAUCHY && !res_out) {  // (GetCost's residuals keep the general form)
      evaluate_partial_c<KC, CFEAR_LOSSK_CAUCHY>(ls, M, lds_match, x0, x1, c, s, nullptr, 0);
    } else {
      evaluate_partial_c<KC, CFEAR_LOSSK_ANY>(ls, M, lds_match, x0, x1, c, s, res_out, res_cap);
    }
    return;
  }
  const int cost = ls->rp.cost;
  // the default configuration (P2L, Huber, matches in LDS) inline in the kernel: out of line, its two interleaved chains reach
  // the callee-saved registers, whose save / restore through scratch is a round trip to memory per evaluation
  if (loss == CFEAR_LOSS_HUBER && cost == CFEAR_COST_P2L && lds_match == 1 && !res_out) {
    evaluate_partial_t<1, CFEAR_COST_P2L, CFEAR_LOSSK_HUBER>(ls, M, x0, x1, c, s, nullptr, 0);
    return;
  }
#define CFEAR_EVAL_BY_COST(LK, RO, RC)                                                                                   \
  do {                                                                                                                  \
    if (cost == CFEAR_COST_P2L) evaluate_partial_c<CFEAR_COST_P2L, LK>(ls, M, lds_match, x0, x1, c, s, RO, RC);         \
    else if (cost == CFEAR_COST_P2D) evaluate_partial_c<CFEAR_COST_P2D, LK>(ls, M, lds_match, x0, x1, c, s, RO, RC);    \
    else evaluate_partial_c<CFEAR_COST_P2P, LK>(ls, M, lds_match, x0, x1, c, s, RO, RC);                                \
  } while (0)
  if (loss == CFEAR_LOSS_HUBER) CFEAR_EVAL_BY_COST(CFEAR_LOSSK_HUBER, res_out, res_cap);
  else if (loss == CFEAR_LOSS_CAUCHY && !res_out) CFEAR_EVAL_BY_COST(CFEAR_LOSSK_CAUCHY, nullptr, 0);
  else CFEAR_EVAL_BY_COST(CFEAR_LOSSK_ANY, res_out, res_cap);
#undef CFEAR_EVAL_BY_COST
}

// W.red is an LDS array: reading it through an LDS-typed pointer gives independent ds_read instructions; through the
// generic pointer every partial sum was a flat load the running sum had to wait for (2.8 us per evaluation).
__device__ __forceinline__ NormalEq gather_partials_regs(const double* red_lds) {
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  lds_cdouble* red = (lds_cdouble*)red_lds;
  double r[10];
#pragma unroll
  for (int h = 0; h < 10; h += 5) {  // five quantities at a time: 20 loads in flight
    double p[5][CFEAR_EVAL_WAVES];
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < CFEAR_EVAL_WAVES; j++) p[i][j] = red[(h + i) * CFEAR_RED_STRIDE + j];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      double t = 0;
#pragma unroll
      for (int j = 0; j < CFEAR_EVAL_WAVES; j++) t += p[i][j];
      r[h + i] = t;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  NormalEq e;  // field order: cost g0 g1 g2 h00 h01 h02 h11 h12 h22
  e.cost = r[0]; e.g0 = r[1]; e.g1 = r[2]; e.g2 = r[3]; e.h00 = r[4]; e.h01 = r[5]; e.h02 = r[6]; e.h11 = r[7]; e.h12 = r[8]; e.h22 = r[9];
  return e;
}
__device__ __forceinline__ void gather_partials(const double* red_lds, LNormalEq* out) {
  typedef __attribute__((address_space(3))) const double lds_cdouble;
  lds_cdouble* red = (lds_cdouble*)red_lds;
  // every registration kernel runs CFEAR_EVAL_WAVES or more waves (static_assert next to BLOCK_R): all slots hold sums
  double r[10];
#pragma unroll
  for (int h = 0; h < 10; h += 5) {  // five quantities at a time: 20 loads in flight, 40 VGPRs
    double p[5][CFEAR_EVAL_WAVES];
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
      for (int j = 0; j < CFEAR_EVAL_WAVES; j++) p[i][j] = red[(h + i) * CFEAR_RED_STRIDE + j];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      double t = 0;
#pragma unroll
      for (int j = 0; j < CFEAR_EVAL_WAVES; j++) t += p[i][j];
      r[h + i] = t;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  typedef __attribute__((address_space(3))) double lds_double;
  lds_double* o = (lds_double*)out;  // field order of NormalEq: cost g0 g1 g2 h00 h01 h02 h11 h12 h22
#pragma unroll
  for (int i = 0; i < 10; i++) o[i] = r[i];  // every lane of the controller wave stores the same values
}

// 3x3 Cholesky solve. The triangular solves multiply by reciprocal square roots of the pivots instead of dividing
// (three rsqrt instead of three sqrt + nine divisions on the controller's serial chain; differs from the oracle's
// division form by a few ulp).
__device__ __forceinline__ bool chol3_solve(const double A[6], const double b[3], double y[3]) {
  const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
  if (!(a00 > 0)) return false;
  const double r0 = rsqrt(a00), l10 = a01 * r0, l20 = a02 * r0;
  const double d1 = a11 - l10 * l10;
  if (!(d1 > 0)) return false;
  const double r1 = rsqrt(d1), l21 = (a12 - l20 * l10) * r1;
  const double d2 = a22 - l20 * l20 - l21 * l21;
  if (!(d2 > 0)) return false;
  const double r2 = rsqrt(d2);
  const double z0 = b[0] * r0, z1 = (b[1] - l10 * z0) * r1, z2 = (b[2] - l20 * z0 - l21 * z1) * r2;
  y[2] = z2 * r2; y[1] = (z1 - l21 * y[2]) * r1; y[0] = (z0 - l10 * y[1] - l20 * y[2]) * r0;
  return isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2]);
}

// the association's view of a cell (ScanDev::rsrc / rtar)
struct RCell { double mx, my, nx, ny, ns, scale; };
__device__ __forceinline__ RCell rcell_src(const ScanDev* S, int j) {  // consecutive lanes read consecutive cells
  const size_t cc = (size_t)S->cap_cells;
  const double* r = S->rsrc + j;
  RCell c; c.mx = r[0]; c.my = r[cc]; c.nx = r[2 * cc]; c.ny = r[3 * cc]; c.ns = r[4 * cc]; c.scale = r[5 * cc];
  return c;
}
__device__ __forceinline__ RCell rcell_tar(const ScanDev* S, int ti) {  // one 64-byte record, three 16-byte loads
  const double2* r = reinterpret_cast<const double2*>(S->rtar + 8 * (size_t)ti);
  const double2 a = r[0], b = r[1], c2 = r[2];
  RCell c; c.mx = a.x; c.my = a.y; c.nx = b.x; c.ny = b.y; c.ns = c2.x; c.scale = c2.y;
  return c;
}

// one match record (AddScanPairCost :266-320) written at position o of the destination SoA
__device__ __forceinline__ void write_match(const MatchPtrs& m, int o, const RegParams& P, const double* T, const double* Tt,
                                            const RCell& cs, const RCell& ct, const double* ct_cov /* xx, xy, yy (P2D) */) {
  const double nx = T[0] * cs.nx + T[1] * cs.ny;
  const double ny = T[2] * cs.nx + T[3] * cs.ny;
  const double sim = fmax(nx * ct.nx + ny * ct.ny, 0.0);
  m.w[o] = get_weight(P.weight_opt, cs.ns, ct.ns, sim, cs.scale, ct.scale);
  m.tmx[o] = (Tt[0] * ct.mx + Tt[1] * ct.my) + Tt[4];
  m.tmy[o] = (Tt[2] * ct.mx + Tt[3] * ct.my) + Tt[5];
  m.sx[o] = cs.mx; m.sy[o] = cs.my;
  if (P.cost == CFEAR_COST_P2D) {  // :290-299
    const double a = ct_cov[0], b = ct_cov[1], c = ct_cov[2];
    const double r00 = Tt[0], r01 = Tt[1], r10 = Tt[2], r11 = Tt[3];
    const double m00 = r00 * a + r01 * b, m01 = r00 * b + r01 * c;
    const double m10 = r10 * a + r11 * b, m11 = r10 * b + r11 * c;
    const double c00 = (P.regularization + (m00 * r00 + m01 * r01)) * P.covar_scale;
    const double c10 = (0.0 + (m10 * r00 + m11 * r01)) * P.covar_scale;
    const double c01 = (0.0 + (m00 * r10 + m01 * r11)) * P.covar_scale;
    const double c11 = (P.regularization + (m10 * r10 + m11 * r11)) * P.covar_scale;
    const double det = c00 * c11 - c01 * c10, id = 1.0 / det;
    const double i00 = c11 * id, i10 = -c10 * id, i11 = c00 * id;
    const double l00 = sqrt(i00), l10 = i10 / l00;
    const double l11 = sqrt(i11 - l10 * l10);
    if (m.a0) m.a0[o] = l00; if (m.a1) m.a1[o] = l10; if (m.a2) m.a2[o] = l11;
  } else {
    if (m.a0) m.a0[o] = Tt[0] * ct.nx + Tt[1] * ct.ny;  // (the LDS layout of a cost leaves out what it never reads)
    if (m.a1) m.a1[o] = Tt[2] * ct.nx + Tt[3] * ct.ny;
    if (m.a2) m.a2[o] = 0;
  }
}

// association of one (keyframe i, source cell j) pair (n_scan_normal.cpp:228-247): index of the matched target
// cell or -1. Out of line: the kernel's register budget is the maximum over its callees.
__device__ __noinline__ int associate_pair(const ScanDev* src, const LRegShared* sh, int nsrc, int p, double curr_radius) {
  const double angle_outlier = 0.86602540378443864676;  // cos(M_PI/6)
  const int i = p / nsrc, j = p - i * nsrc;
  const auto* T = sh->Trel[i];
  const size_t cc = (size_t)src->cap_cells;
  const double* rs = src->rsrc + j;
  const double mx = rs[0], my = rs[cc];
  const double snx = rs[2 * cc], sny = rs[3 * cc];
  const double qx = (T[0] * mx + T[1] * my) + T[4];
  const double qy = (T[2] * mx + T[3] * my) + T[5];
  int ti = scan_closest(CFEAR_GENERIC(const GridView, sh->kf[i]), qx, qy, curr_radius);
  if (ti >= 0) {
    const double2 tn = reinterpret_cast<const double2*>(sh->kf[i].rtar + 8 * (size_t)ti)[1];
    const double nx = T[0] * snx + T[1] * sny;
    const double ny = T[2] * snx + T[3] * sny;
    const double sim = fmax(nx * tn.x + ny * tn.y, 0.0);
    if (!(sim > angle_outlier)) ti = -1;  // :247
  }
  return ti;
}
// ... under a tie rule other than the production one (RegParams::nn_tie): the same pair through scan_closest_rule; the kd descent's stack
// is this thread's slice of the match arrays in memory (free until the emission that follows every association of the general path)
__device__ __noinline__ int associate_pair_rule(ScanDev* const* scans, const ScanDev* src, const LRegShared* sh, int nsrc, int p, double curr_radius, int rule) {
  const double angle_outlier = 0.86602540378443864676;  // cos(M_PI/6)
  const int i = p / nsrc, j = p - i * nsrc;
  const auto* T = sh->Trel[i];
  const size_t cc = (size_t)src->cap_cells;
  const double* rs = src->rsrc + j;
  const double mx = rs[0], my = rs[cc];
  const double snx = rs[2 * cc], sny = rs[3 * cc];
  const double qx = (T[0] * mx + T[1] * my) + T[4];
  const double qy = (T[2] * mx + T[3] * my) + T[5];
  KdVisit* stack = reinterpret_cast<KdVisit*>(sh->rw.tmx) + (size_t)threadIdx.x * CFEAR_KD_STACK;
  int ti = scan_closest_rule(scans[i], CFEAR_GENERIC(const GridView, sh->kf[i]), qx, qy, curr_radius, rule, stack);
  if (ti >= 0) {
    const double2 tn = reinterpret_cast<const double2*>(sh->kf[i].rtar + 8 * (size_t)ti)[1];
    const double nx = T[0] * snx + T[1] * sny;
    const double ny = T[2] * snx + T[3] * sny;
    const double sim = fmax(nx * tn.x + ny * tn.y, 0.0);
    if (!(sim > angle_outlier)) ti = -1;  // :247
  }
  return ti;
}
__device__ __noinline__ void emit_match(ScanDev* const* scans, const ScanDev* src, const LRegShared* sh, int nsrc, int p, int ti, int o, bool use_lds) {
  const RegParams& P = CFEAR_GENERIC(const RegParams, sh->rp);
  const RegScratch& W = CFEAR_GENERIC(const RegScratch, sh->rw);
  const int i = p / nsrc, j = p - i * nsrc;
  const RCell cs = rcell_src(src, j);
  RCell ct;
  {
    const double2* r = reinterpret_cast<const double2*>(sh->kf[i].rtar + 8 * (size_t)ti);  // the LDS view: no pointer chase
    const double2 r0 = r[0], r1 = r[1], r2 = r[2];
    ct.mx = r0.x; ct.my = r0.y; ct.nx = r1.x; ct.ny = r1.y; ct.ns = r2.x; ct.scale = r2.y;
  }
  const double* ctf = (P.cost == CFEAR_COST_P2D) ? scans[i]->rcov + 3 * (size_t)ti : nullptr;  // the target's covariance xx, xy, yy
  const double* Trel = (const double*)sh->Trel[i];  // generic views for the by-pointer interface of write_match
  const double* Ttar = (const double*)sh->Ttar[i];
  if (use_lds) write_match(match_ptrs_lds(P.cost), o, P, Trel, Ttar, cs, ct, ctf);
  else write_match(match_ptrs(W.tmx, (size_t)W.cap), o, P, Trel, Ttar, cs, ct, ctf);
}

// ---- association of one source cell against up to four keyframes ------------------------------------------------
// A pair's search is a chain of dependent memory round trips (source cell -> bucket bounds -> candidates -> target
// normal), about 1 us each; pair after pair that chain was the whole cost of the association.
struct Assoc4 { int t0, t1, t2, t3; };
__device__ __forceinline__ int assoc_get(const Assoc4& a, int i) { return i == 0 ? a.t0 : (i == 1 ? a.t1 : (i == 2 ? a.t2 : a.t3)); }

// All (up to four) keyframes of a source cell in one call, staged: the source cell (one round trip), the bucket bounds of
// ALL keyframes (one), then keyframes 0, 1: candidates (NC per keyframe and trip) and their gate normals - whose round trip
// overlaps the first candidate trip of keyframes 2, 3 - then those. About three round trips less per source cell than a
// two-keyframe search called twice (source cell, bucket bounds, a gate), with the register footprint of a two-keyframe search
// plus the parked bounds of the second pair (all four keyframes side by side did not fit the registers). Same results as
// scan_closest + the gate of associate_pair.
// Branch-free with clamped addresses and global-typed pointers: every load of a phase is issued unconditionally (a keyframe
// that does not exist, an empty window or a candidate past the end reads a valid dummy address and is masked afterwards), so
// that the loads of a phase are in flight together - with the loads inside conditionals the compiler waited for one keyframe's
// data before it issued the next one's, and generic pointers made them flat loads that also wait for the LDS counter. The
// candidate update uses selects and non-short-circuit predicates: a load whose only use sits in a conditional block gets sunk
// into it and waited for alone. Returns the four matches 16 bits each (0xFFFF = none, 0xFFFE = left to
// the caller), which needs <= 65533 cells per keyframe (else that keyframe is left to the caller).
// kbase, nk: the (up to four) keyframes kbase .. kbase + nk - 1 of the registration (a submap of more than four keyframes is
// searched in groups of four)
__device__ __noinline__ unsigned long long associate_cell4(const LRegShared* sh, int k0, int nk, int j, double curr_radius) {
  typedef __attribute__((address_space(1))) const double g_cf64;
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) const u32x2 g_cu32x2;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(1))) const f32x4 g_cf32x4;
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) const f64x2 g_cf64x2;
  constexpr int NC = 4;
  const double angle_outlier = 0.86602540378443864676;  // cos(M_PI/6)
  const size_t cc = (size_t)sh->scc;
  g_cf64* rs = (g_cf64*)sh->srs + j;
  const double mx = rs[0], my = rs[cc], snx = rs[2 * cc], sny = rs[3 * cc];
  const double m = curr_radius * (1.0 + 1e-6) + 1e-6;  // window padding of scan_closest
  float qx[4], qy[4];
  int nrows[4];
  unsigned wide = 0;
  u32x2 L[4], H[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {  // windows and bucket bounds of all keyframes: one round trip
    const int i = k0 + min(u, nk - 1);
    const auto* T = sh->Trel[i];
    qx[u] = (float)((T[0] * mx + T[1] * my) + T[4]);
    qy[u] = (float)((T[2] * mx + T[3] * my) + T[5]);
    const int gw = sh->kf[i].gw, gh = sh->kf[i].gh, nc = sh->kf[i].n_cells;
    const double igc = sh->kf[i].igc, gmx = (double)sh->kf[i].gminx, gmy = (double)sh->kf[i].gminy;
    int ax0 = (int)floor(((double)qx[u] - m - gmx) * igc), ax1 = (int)floor(((double)qx[u] + m - gmx) * igc);
    int ay0 = (int)floor(((double)qy[u] - m - gmy) * igc), ay1 = (int)floor(((double)qy[u] + m - gmy) * igc);
    ax0 = max(ax0, 0); ay0 = max(ay0, 0); ax1 = min(ax1, gw - 1); ay1 = min(ay1, gh - 1);
    const bool ok = (u < nk) && nc > 0 && gw > 0 && ax0 <= ax1 && ay0 <= ay1;
    const bool wd = ok && (ay1 - ay0 > 2 || nc > CFEAR_GRID16_MAX);  // more than three rows of buckets, or a scan without 16-bit offsets: the caller's
    wide |= wd ? (1u << u) : 0u;
    const bool use = ok && !wd;
    nrows[u] = use ? ay1 - ay0 + 1 : 0;
    const int b0 = use ? ay0 * gw + ax0 : 0, b1 = use ? ay0 * gw + ax1 + 1 : 0;
    // the offsets of the window's (up to) three bucket rows: 16-bit values b, b + gw, b + 2 gw apart (grid_off16: padded behind the
    // last bucket, so no clamping), two to a register
    typedef __attribute__((address_space(1))) const unsigned short g_cu16;
    g_cu16* o16 = (g_cu16*)grid_off16(sh->kf[i].gs);
    const int gw2 = use ? gw : 0;
    L[u].x = (unsigned)o16[b0] | ((unsigned)o16[b0 + gw2] << 16); L[u].y = (unsigned)o16[b0 + 2 * gw2];
    H[u].x = (unsigned)o16[b1] | ((unsigned)o16[b1 + gw2] << 16); H[u].y = (unsigned)o16[b1 + 2 * gw2];
  }
  int ti[4];
  f64x2 tn[2];
  auto search = [&](int u0) {  // keyframes u0, u0 + 1: nearest candidate of each
    int lo[2][3], cnt[2][2], tot[2], kmax = 0;
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const int u = u0 + v;
      const int l0 = (int)(L[u].x & 0xFFFFu), l1 = (int)(L[u].x >> 16), l2 = (int)L[u].y;
      const int h0 = (int)(H[u].x & 0xFFFFu), h1 = (int)(H[u].x >> 16), h2 = (int)H[u].y;
      const int c0 = (0 < nrows[u]) ? h0 - l0 : 0;
      const int c1 = (1 < nrows[u]) ? h1 - l1 : 0;
      const int c2 = (2 < nrows[u]) ? h2 - l2 : 0;
      cnt[v][0] = c0; cnt[v][1] = c0 + c1; tot[v] = c0 + c1 + c2;
      lo[v][0] = l0; lo[v][1] = l1 - c0; lo[v][2] = l2 - (c0 + c1);
      kmax = max(kmax, tot[v]);
    }
    // nearest candidate, exact-distance ties to the lowest cell index: the minimum of the 64-bit keys (distance bits << 32 | cell
    // index) - a squared distance is never negative, so its bit pattern orders like its value - one 64-bit compare and two
    // selects per candidate where the two-level comparison took nine instructions
    unsigned long long bk[2] = {~0ull, ~0ull};
    for (int k = 0; k < kmax; k += NC) {
      f32x4 c[2][NC];
#pragma unroll
      for (int v = 0; v < 2; v++) {
        g_cf32x4* gp = (g_cf32x4*)sh->kf[k0 + min(u0 + v, nk - 1)].gp;
#pragma unroll
        for (int w = 0; w < NC; w++) {
          const int kk = k + w;
          const int idx = kk + (kk < cnt[v][0] ? lo[v][0] : (kk < cnt[v][1] ? lo[v][1] : lo[v][2]));
          c[v][w] = gp[kk < tot[v] ? idx : 0];
        }
      }
#pragma unroll
      for (int v = 0; v < 2; v++) {
#pragma unroll
        for (int w = 0; w < NC; w++) {
          const float dx = qx[u0 + v] - c[v][w].x, dy = qy[u0 + v] - c[v][w].y;
          float d2 = dx * dx; d2 += dy * dy;
          const unsigned long long key = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)__float_as_uint(c[v][w].z);
          const bool take = (k + w < tot[v]) & (key < bk[v]);
          bk[v] = take ? key : bk[v];
        }
      }
    }
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const int u = u0 + v;
      const float bd = __uint_as_float((unsigned)(bk[v] >> 32));
      ti[u] = (bk[v] != ~0ull && (double)bd < curr_radius * curr_radius) ? (int)(unsigned)bk[v] : -1;
      if (u >= nk) ti[u] = -1;
    }
  };
  auto gate_load = [&](int u0) {
#pragma unroll
    for (int v = 0; v < 2; v++) tn[v] = ((g_cf64x2*)(sh->kf[k0 + min(u0 + v, nk - 1)].rtar + 8 * (size_t)(ti[u0 + v] >= 0 ? ti[u0 + v] : 0)))[1];
  };
  auto gate = [&](int u0) {
#pragma unroll
    for (int v = 0; v < 2; v++) {
      const int u = u0 + v;
      const auto* T = sh->Trel[k0 + min(u, nk - 1)];
      const double nx = T[0] * snx + T[1] * sny;
      const double ny = T[2] * snx + T[3] * sny;
      const double sim = fmax(nx * tn[v].x + ny * tn[v].y, 0.0);
      if (ti[u] >= 0 && !(sim > angle_outlier)) ti[u] = -1;  // :247
      if ((wide & (1u << u)) && u < nk) ti[u] = -2;
    }
  };
  search(0);
  gate_load(0);
  if (nk > 2) {  // block-uniform
    search(2);   // (the gate normals of keyframes 0, 1 arrive during the first candidate trip)
    gate(0);
    gate_load(2);
    gate(2);
  } else {
    gate(0);
    ti[2] = ti[3] = -1;
  }
  unsigned long long r = 0;
#pragma unroll
  for (int u = 0; u < 4; u++) r |= (unsigned long long)(unsigned)(ti[u] & 0xFFFF) << (16 * u);
  return r;
}
__device__ __forceinline__ Assoc4 associate_cell(const ScanDev* src, const LRegShared* sh, int k0, int nk, int j, double curr_radius) {
  Assoc4 a;
  {
    const unsigned long long p = associate_cell4(sh, k0, nk, j, curr_radius);
    auto un = [](unsigned v) -> int { return v >= 0xFFFEu ? (int)v - 0x10000 : (int)v; };  // 0xFFFF -> -1, 0xFFFE -> -2
    a.t0 = un((unsigned)(p & 0xFFFF)); a.t1 = un((unsigned)((p >> 16) & 0xFFFF)); a.t2 = un((unsigned)((p >> 32) & 0xFFFF)); a.t3 = un((unsigned)(p >> 48));
  }
  if (a.t0 == -2 || a.t1 == -2 || a.t2 == -2 || a.t3 == -2) {  // rare: the general search for those pairs
    const int nsrc = src->n_cells;
    if (a.t0 == -2) a.t0 = associate_pair(src, sh, nsrc, (k0 + 0) * nsrc + j, curr_radius);
    if (a.t1 == -2) a.t1 = associate_pair(src, sh, nsrc, (k0 + 1) * nsrc + j, curr_radius);
    if (a.t2 == -2) a.t2 = associate_pair(src, sh, nsrc, (k0 + 2) * nsrc + j, curr_radius);
    if (a.t3 == -2) a.t3 = associate_pair(src, sh, nsrc, (k0 + 3) * nsrc + j, curr_radius);
  }
  return a;
}

// residual blocks of one source cell (up to four keyframes); pos = four 16-bit positions in the match arrays.
// The source cell is read once; matches that fit the LDS array are stored through an LDS-typed pointer (ds_write, not
// flat stores through the address unit). Same arithmetic as write_match.
template <int KCOST = -1>
__device__ __forceinline__ void emit_cell(ScanDev* const* scans, const ScanDev* src, const LRegShared* sh,
                                          int nsrc, int k0, int nk, int j, Assoc4 a, unsigned long long pos, int mode /* RegShared::lds_match */) {
  // block-uniform values in scalar registers: read from LDS they sit in vector registers, and every branch on them is compiled as
  // a divergent one (save the exec mask, branch, restore)
  const int cost = KCOST >= 0 ? KCOST : __builtin_amdgcn_readfirstlane(sh->rp.cost), weight_opt = __builtin_amdgcn_readfirstlane(sh->rp.weight_opt);
  mode = __builtin_amdgcn_readfirstlane(mode);
  typedef __attribute__((address_space(1))) const double g_cf64;
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(1))) const f64x2 g_cf64x2;
  RCell cs;
  {  // the source cell through the pointer kept in LDS (no read of the scan header first), global_load instead of flat
    const size_t cc = (size_t)sh->scc;
    g_cf64* r = (g_cf64*)sh->srs + j;
    cs.mx = r[0]; cs.my = r[cc]; cs.nx = r[2 * cc]; cs.ny = r[3 * cc]; cs.ns = r[4 * cc]; cs.scale = r[5 * cc];
  }
  (void)src;
  typedef __attribute__((address_space(3))) double lds_double;
  // the target records of two keyframes are fetched together (the first pair with the source cell): two round trips to
  // memory for the emission (keyframe by keyframe it was one each plus the source's; all four at once needs the callee-saved
  // registers, whose save / restore through scratch is a round trip of its own). Unmatched / missing keyframes read record 0.
  f64x2 R0[2], R1[2], R2[2];
#pragma unroll
  for (int i = 0; i < 4; i++) {  // branch-free (stores predicated): the pair's independent chains of double-precision arithmetic interleave
    if ((i & 1) == 0) {
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 2; u++) {
        const int tu = assoc_get(a, i + u);
        g_cf64x2* r = (g_cf64x2*)(sh->kf[k0 + min(i + u, nk - 1)].rtar + 8 * (size_t)(tu >= 0 ? tu : 0));
        R0[u] = r[0]; R1[u] = r[1]; R2[u] = r[2];  // mean, normal, (samples, scale)
      }
    }
    const int ti = assoc_get(a, i);
    const bool on = ti >= 0 && i < nk;
    const int tix = on ? ti : 0;
    const int ki = k0 + min(i, nk - 1);
    const int o = (int)((pos >> (16 * i)) & 0xFFFF);
    lds_double* lm = (lds_double*)lds_match_base() + o;
    double* gm = sh->rw.tmx + o;
    const size_t gcap = (size_t)sh->rw.cap;
    const f64x2 r0 = R0[i & 1], r1 = R1[i & 1], r2 = R2[i & 1];
    const auto* T = sh->Trel[ki];
    const auto* Tt = sh->Ttar[ki];
    const double tmx = (Tt[0] * r0.x + Tt[1] * r0.y) + Tt[4];
    const double tmy = (Tt[2] * r0.x + Tt[3] * r0.y) + Tt[5];
    const double nx = T[0] * cs.nx + T[1] * cs.ny;
    const double ny = T[2] * cs.nx + T[3] * cs.ny;
    const double sim = fmax(nx * r1.x + ny * r1.y, 0.0);
    const double wgt = get_weight(weight_opt, cs.ns, r2.x, sim, cs.scale, r2.y);
    double a0, a1, a2;
    if (cost == CFEAR_COST_P2D) {  // :290-299
      const double* ctf = scans[ki]->rcov + 3 * (size_t)tix;
      const double ca = ctf[0], cb = ctf[1], cc = ctf[2];
      const double r00 = Tt[0], r01 = Tt[1], r10 = Tt[2], r11 = Tt[3];
      const double m00 = r00 * ca + r01 * cb, m01 = r00 * cb + r01 * cc;
      const double m10 = r10 * ca + r11 * cb, m11 = r10 * cb + r11 * cc;
      const double c00 = (sh->rp.regularization + (m00 * r00 + m01 * r01)) * sh->rp.covar_scale;
      const double c10 = (0.0 + (m10 * r00 + m11 * r01)) * sh->rp.covar_scale;
      const double c01 = (0.0 + (m00 * r10 + m01 * r11)) * sh->rp.covar_scale;
      const double c11 = (sh->rp.regularization + (m10 * r10 + m11 * r11)) * sh->rp.covar_scale;
      const double det = c00 * c11 - c01 * c10, id = 1.0 / det;
      const double i00 = c11 * id, i10 = -c10 * id, i11 = c00 * id;
      const double l00 = sqrt(i00), l10 = i10 / l00;
      a0 = l00; a1 = l10; a2 = sqrt(i11 - l10 * l10);
    } else {
      a0 = Tt[0] * r1.x + Tt[1] * r1.y;
      a1 = Tt[2] * r1.x + Tt[3] * r1.y;
      a2 = 0.0;
    }
    if (on) {  // one predicated region for the stores of a match; arrays in the order of match_ptrs(): tmx tmy a0 a1 a2 sx sy w
      const int lc = match_lds_cap(cost);
      if (mode == 1 || (mode == 2 && o < lc)) {  // the LDS array keeps what the cost reads (match_lds_idx)
        lm[0] = tmx; lm[lc] = tmy;
        if (cost == CFEAR_COST_P2D) { lm[2 * lc] = a0; lm[3 * lc] = a1; lm[4 * lc] = a2; lm[5 * lc] = cs.mx; lm[6 * lc] = cs.my; lm[7 * lc] = wgt; }
        else if (cost == CFEAR_COST_P2L) { lm[2 * lc] = a0; lm[3 * lc] = a1; lm[4 * lc] = cs.mx; lm[5 * lc] = cs.my; lm[6 * lc] = wgt; }
        else { lm[2 * lc] = cs.mx; lm[3 * lc] = cs.my; lm[4 * lc] = wgt; }
      } else {
        gm[0] = tmx; gm[gcap] = tmy; gm[2 * gcap] = a0; gm[3 * gcap] = a1; gm[4 * gcap] = a2; gm[5 * gcap] = cs.mx; gm[6 * gcap] = cs.my; gm[7 * gcap] = wgt;
      }
    }
  }
}

// The fast path works on blocks of blockDim source cells: thread <-> source cell, its keyframes searched together; one
// packed scan of four 16-bit counters per block orders the matches as the reference does (pair index i * nsrc + j
// ascending). The per-block steps are straight-line functions of their own: a value that lives across a call sits in a
// register the callee does not touch, above the callee's own, so every such value raises the kernel's register count.
struct AssocBlock { unsigned long long e, tb; Assoc4 a; };  // exclusive positions / totals per keyframe (16-bit fields), matches
__device__ __forceinline__ unsigned long long assoc_counts(const Assoc4& a) {
  return (unsigned long long)(a.t0 >= 0) | ((unsigned long long)(a.t1 >= 0) << 16) | ((unsigned long long)(a.t2 >= 0) << 32) |
         ((unsigned long long)(a.t3 >= 0) << 48);
}
// block b of the source cells against the keyframes k0 .. k0 + nk - 1 (group g of four); park: the matches wait in W.assoc (slot
// g * nsrc + j) for the totals of all blocks and groups
__device__ __forceinline__ AssocBlock assoc_block(const ScanDev* src, const LRegShared* sh, int k0, int nk, int nsrc, int itr, int b, int g, bool park) {
  AssocBlock R;
  R.a.t0 = R.a.t1 = R.a.t2 = R.a.t3 = -1;
  const int j = b * CFEAR_REG_BLOCK + threadIdx.x;
  if (j < nsrc) {
    const double curr_radius = (itr == 1) ? 2 * sh->rp.assoc_radius : sh->rp.assoc_radius;  // :222
    R.a = associate_cell(src, sh, k0, nk, j, curr_radius);
  }
  R.e = block_exclusive_scan64<CFEAR_REG_BLOCK>(assoc_counts(R.a), reinterpret_cast<unsigned long long*>(sh->rw.red), &R.tb);
  if (park && j < nsrc) reinterpret_cast<int4*>(sh->rw.assoc)[(size_t)g * nsrc + j] = make_int4(R.a.t0, R.a.t1, R.a.t2, R.a.t3);
  return R;
}
// residual blocks of block b of the source cells for the keyframes of group g; before = positions the group's keyframes start at
// plus their matches in earlier blocks. Returns the matches of the block per keyframe (0 when nothing was parked: nothing follows).
template <int KCOST = -1>
__device__ __noinline__ unsigned long long emit_block(ScanDev* const* scans, const ScanDev* src, const LRegShared* sh, int k0, int nk, int nsrc, int b, int g,
                                                      bool parked, Assoc4 a, unsigned long long e, unsigned long long before, int mode) {
  const int j = b * CFEAR_REG_BLOCK + threadIdx.x;
  unsigned long long tb = 0;
  if (parked) {  // positions inside the block: the same scan again (cheaper than keeping them)
    a.t0 = a.t1 = a.t2 = a.t3 = -1;
    if (j < nsrc) { const int4 v = reinterpret_cast<const int4*>(sh->rw.assoc)[(size_t)g * nsrc + j]; a.t0 = v.x; a.t1 = v.y; a.t2 = v.z; a.t3 = v.w; }
    e = block_exclusive_scan64<CFEAR_REG_BLOCK>(assoc_counts(a), reinterpret_cast<unsigned long long*>(sh->rw.red), &tb);
  }
  if (a.t0 >= 0 || a.t1 >= 0 || a.t2 >= 0 || a.t3 >= 0) emit_cell<KCOST>(scans, src, sh, nsrc, k0, nk, j, a, before + e, mode);
  return tb;
}

// the residual blocks of one (group of four keyframes, source cell) item of the grouped path; out of line like emit_block (inlined, the
// emission's registers are the kernel's)
template <int KCOST = -1>
__device__ __noinline__ void emit_item(ScanDev* const* scans, const ScanDev* src, const LRegShared* sh, int nsrc, int k0, int nk, int j, int4 v,
                                       unsigned long long pos, int mode) {
  const Assoc4 a = {v.x, v.y, v.z, v.w};
  emit_cell<KCOST>(scans, src, sh, nsrc, k0, nk, j, a, pos, mode);
}

template <int KCOST = -1>
__device__ __forceinline__ int build_problem_block(ScanDev* const* scans, int n, LRegShared* sh, int itr) {
  const ScanDev* src = scans[n - 1];
  const int nsrc = sh->kf[n - 1].n_cells;  // (the view in LDS: src->n_cells is a round trip to memory in front of every association)
  const int pairs = (n - 1) * nsrc;
  const int nt = CFEAR_REG_BLOCK, tid = threadIdx.x;
  int M;
  int mode;  // RegShared::lds_match
  const int nk = n - 1;
  const int ngroups = (nk + 3) >> 2;
  const int lcap = match_lds_cap(KCOST >= 0 ? KCOST : sh->rp.cost);
  // the grouped path parks four ints + two ints of positions per (group, source cell) in W.assoc
  const bool can_park = 6 * (long long)ngroups * nsrc <= (long long)sh->rw.acap && (reinterpret_cast<uintptr_t>(sh->rw.assoc) & 15) == 0;
  bool done = false;
  int assoc_path = 3;
  const int tie_rule = sh->rp.nn_tie;  // (block-uniform) a non-production tie rule: the general path below
  if (tie_rule == 0 && nk <= 4 && nsrc <= nt) {  // one group of keyframes, one block of cells: the matches stay in registers
    const AssocBlock R = assoc_block(src, sh, 0, nk, nsrc, itr, 0, 0, false);
    const unsigned long long T = R.tb;  // matches per keyframe, 16-bit fields
    const unsigned long long t0 = T & 0xFFFF, t1 = (T >> 16) & 0xFFFF, t2 = (T >> 32) & 0xFFFF, t3 = (T >> 48) & 0xFFFF;
    M = (int)(t0 + t1 + t2 + t3);
    mode = M <= lcap ? 1 : 2;
    (void)emit_block<KCOST>(scans, src, sh, 0, nk, nsrc, 0, 0, false, R.a, R.e, (t0 << 16) | ((t0 + t1) << 32) | ((t0 + t1 + t2) << 48), mode);
    done = true; assoc_path = 1;
  } else if (tie_rule == 0 && can_park && (long long)nk * nsrc <= 65535 && nk <= 64) {
    // several blocks of cells and / or several groups of four keyframes (a submap of 5 .. 63 keyframes: the reference's s10 and s50
    // presets; a dense scan against four): the items (group g of four keyframes, source cell j), numbered g * nsrc + j, are dealt to
    // the threads DENSELY - with 172 source cells and 13 groups a 512-thread workgroup makes 5 passes where group after group it made
    // 13, each a chain of dependent memory round trips. A pass searches with the four-keyframes-at-once association and parks the
    // matches; then one wave per group counts (ballots, no barrier) and leaves every item's position inside its keyframes; the residual
    // blocks are numbered as the reference does - pair index keyframe * nsrc + cell ascending - from the per-keyframe totals (16-bit
    // fields: at most 65535 residual blocks); the emission walks the items densely again.
    const double curr_radius = (itr == 1) ? 2 * sh->rp.assoc_radius : sh->rp.assoc_radius;  // :222
    const int nitems = ngroups * nsrc;
    int4* park = reinterpret_cast<int4*>(sh->rw.assoc);
    unsigned long long* ppos = reinterpret_cast<unsigned long long*>(sh->rw.assoc + 4 * (size_t)nitems);  // (16-byte aligned base + 16 * nitems: 8-byte aligned)
    unsigned long long* gt = reinterpret_cast<unsigned long long*>(sh->rw.red_i);  // [0..15] totals per group, [16..31] where a group's keyframes start
    for (int it = tid; it < nitems; it += nt) {
      const int g = it / nsrc, j = it - g * nsrc;
      const Assoc4 a = associate_cell(src, sh, 4 * g, min(4, nk - 4 * g), j, curr_radius);
      park[it] = make_int4(a.t0, a.t1, a.t2, a.t3);
    }
    __syncthreads();  // every parked match is visible
    {  // wave w counts the groups w, w + waves, ...: positions of a group's matches per keyframe, in cell order
      const int lane = lane_id(), wv = tid >> 6, nwv = nt >> 6;
      for (int g = wv; g < ngroups; g += nwv) {
        unsigned long long run = 0;  // matches so far per keyframe, 16-bit fields
        for (int j0 = 0; j0 < nsrc; j0 += 64) {
          const int j = j0 + lane;
          const bool in = j < nsrc;
          const int4 v = in ? park[(size_t)g * nsrc + j] : make_int4(-1, -1, -1, -1);
          const unsigned long long b0 = __ballot(v.x >= 0), b1 = __ballot(v.y >= 0), b2 = __ballot(v.z >= 0), b3 = __ballot(v.w >= 0);
          const unsigned long long below = (1ull << lane) - 1ull;
          const unsigned long long e = (unsigned long long)__popcll(b0 & below) | ((unsigned long long)__popcll(b1 & below) << 16) |
                                       ((unsigned long long)__popcll(b2 & below) << 32) | ((unsigned long long)__popcll(b3 & below) << 48);
          if (in) ppos[(size_t)g * nsrc + j] = run + e;
          run += (unsigned long long)__popcll(b0) | ((unsigned long long)__popcll(b1) << 16) | ((unsigned long long)__popcll(b2) << 32) |
                 ((unsigned long long)__popcll(b3) << 48);
        }
        if (lane == 0) gt[g] = run;
      }
    }
    __syncthreads();
    if (tid == 0) {  // where the keyframes of every group start (a handful of groups: serial)
      unsigned long long base = 0;
      for (int g = 0; g < ngroups; g++) {
        const unsigned long long T = gt[g];
        const unsigned long long t0 = T & 0xFFFF, t1 = (T >> 16) & 0xFFFF, t2 = (T >> 32) & 0xFFFF, t3 = (T >> 48) & 0xFFFF;
        gt[16 + g] = base | ((base + t0) << 16) | ((base + t0 + t1) << 32) | ((base + t0 + t1 + t2) << 48);
        base += t0 + t1 + t2 + t3;
      }
      gt[0] = base;  // (the totals have been consumed)
    }
    __syncthreads();
    M = (int)gt[0];
    mode = M <= lcap ? 1 : 2;
    for (int it = tid; it < nitems; it += nt) {
      const int g = it / nsrc, j = it - g * nsrc;
      const int4 v = park[it];
      if (v.x >= 0 || v.y >= 0 || v.z >= 0 || v.w >= 0) emit_item<KCOST>(scans, src, sh, nsrc, 4 * g, min(4, nk - 4 * g), j, v, gt[16 + g] + ppos[it], mode);
    }
    done = true; assoc_path = 2;
  }
  if (!done) {  // anything else (thousands of cells per scan): contiguous pair ranges per thread, associations parked in global memory
    const double curr_radius = (itr == 1) ? 2 * sh->rp.assoc_radius : sh->rp.assoc_radius;  // :222
    const int ipt = (pairs + nt - 1) / nt;
    const int p0 = tid * ipt, p1 = min(pairs, p0 + ipt);
    int cnt = 0;
    for (int p = p0; p < p1; p++) {
      // (the kd descent's per-thread stack needs CFEAR_KD_STACK * 16 B * blockDim of the match arrays: 64 B per pair of capacity)
      // A parity mode that cannot be honoured is an error, not the production rule in disguise: assoc_path goes out negative and the host entry points refuse
      // the result (CFEAR_ERR_UNSUPPORTED).
      const bool rule_ok = tie_rule == 0 || (size_t)sh->rw.cap * 64 >= (size_t)nt * CFEAR_KD_STACK * sizeof(KdVisit);
      if (!rule_ok) assoc_path = -3;
      const int ti = (tie_rule != 0 && rule_ok) ? associate_pair_rule(scans, src, sh, nsrc, p, curr_radius, tie_rule) : associate_pair(src, sh, nsrc, p, curr_radius);
      sh->rw.assoc[p] = ti;
      cnt += (ti >= 0) ? 1 : 0;
    }
    int o = block_exclusive_scan<CFEAR_REG_BLOCK>(cnt, sh->rw.red_i, &M);
    // more residual blocks than the LDS array holds: its capacity stays in LDS, the rest goes to memory (mode 2, as on the fast
    // paths) - every evaluation of a solve reads the matches again
    mode = M <= lcap ? 1 : 2;
    for (int p = p0; p < p1; p++) {
      const int ti = sh->rw.assoc[p];
      if (ti >= 0) { emit_match(scans, src, sh, nsrc, p, ti, o, mode == 1 || o < lcap); o++; }
    }
  }
  if (tid == 0) { sh->lds_match = mode; sh->assoc_path = assoc_path; }
  __syncthreads();
  return M;
}

// ---------------------------------------------------------------------------------------------
// Controller (wave 0 only, all 64 lanes redundantly): a state machine advanced once per command.
// Same arithmetic, in the same order, as the CPU oracle's register / LM routines (tests compare iteration counts).
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ void ctl_publish_eval(LRegShared* sh, double x0, double x1, double x2, int state, double cs, double sn) {
  sh->x[0] = x0; sh->x[1] = x1; sh->x[2] = x2;
  sh->c = cs; sh->s = sn;
  sh->cmd = REG_CMD_EVAL; sh->state = state;
}
__device__ __forceinline__ void ctl_publish_eval_cur(LRegShared* sh, int state) {  // at xcur: its cos / sin are known
  ctl_publish_eval(sh, sh->xcur[0], sh->xcur[1], sh->xcur[2], state, sh->cur_c, sh->cur_s);
}

__device__ __noinline__ void ctl_publish_candidate(LRegShared* sh) {
  const double x2 = sh->xc[2];
  double sn, cs;
  sincos(x2, &sn, &cs);
  ctl_publish_eval(sh, sh->xc[0], sh->xc[1], x2, REG_ST_LM_CAND, cs, sn);
}

// transforms of all keyframes for the current pose of the last scan; lane i handles keyframe i. first: the first association
// of a registration computes the keyframe maps and the cos / sin of the pose; the later ones find the maps in LDS (the
// keyframes do not move) and the cos / sin where the evaluation that led to xcur left them - the same numbers, without four
// sincos calls on the controller's serial chain per re-association
__device__ __noinline__ void ctl_publish_build(LRegShared* sh, bool first) {
  const auto& io = sh->rio;  // fields read through the LDS-typed pointer
  const int n = io.n, L = 3 * (n - 1);
  Aff2 Tsrc;
  if (first) {
    Tsrc = aff_from_xyt(sh->xcur[0], sh->xcur[1], sh->xcur[2]);
    sh->cur_c = Tsrc.l0; sh->cur_s = Tsrc.l2; sh->prev_c = Tsrc.l0; sh->prev_s = Tsrc.l2;
  } else {
    const double cs = sh->cur_c, sn = sh->cur_s;
    Tsrc.l0 = cs; Tsrc.l1 = -sn; Tsrc.l2 = sn; Tsrc.l3 = cs; Tsrc.t0 = sh->xcur[0]; Tsrc.t1 = sh->xcur[1];
  }
  for (int i = lane_id(); i < n - 1; i += 64) {
    auto* a = sh->Ttar[i]; auto* b = sh->Trel[i];
    Aff2 Tt;
    if (first) {
      Tt = aff_from_xyt(io.par[3 * i], io.par[3 * i + 1], io.par[3 * i + 2]);
      a[0] = Tt.l0; a[1] = Tt.l1; a[2] = Tt.l2; a[3] = Tt.l3; a[4] = Tt.t0; a[5] = Tt.t1;
    } else {
      Tt.l0 = a[0]; Tt.l1 = a[1]; Tt.l2 = a[2]; Tt.l3 = a[3]; Tt.t0 = a[4]; Tt.t1 = a[5];
    }
    const Aff2 Tr = aff_mul(aff_inv(Tt), Tsrc);  // Tsrctotar (:224)
    b[0] = Tr.l0; b[1] = Tr.l1; b[2] = Tr.l2; b[3] = Tr.l3; b[4] = Tr.t0; b[5] = Tr.t1;
  }
  io.par[L] = sh->xcur[0]; io.par[L + 1] = sh->xcur[1]; io.par[L + 2] = sh->xcur[2];
  sh->x[0] = sh->xcur[0]; sh->x[1] = sh->xcur[1]; sh->x[2] = sh->xcur[2];
  sh->c = Tsrc.l0; sh->s = Tsrc.l2;  // the build command ends with the evaluation at x (the LM's iteration 0)
  sh->cmd = REG_CMD_BUILD; sh->state = REG_ST_BUILD;
}

__device__ __noinline__ void ctl_finish(LRegShared* sh, bool have_cov, const LNormalEq* Ep /* sh->E or sh->G; unused without have_cov */) {
  const NormalEq E = neq_load(Ep);
  const auto& io = sh->rio;
  const int n = io.n, L = 3 * (n - 1), lane = lane_id();
  int ret = 0;
  // every lane of the controller wave computes the same values; the stores are spread over the lanes
  if (sh->success) {
    for (int i = lane; i < L; i += 64) io.poses[i] = io.par[i];
    if (lane < 3) io.poses[L + lane] = sh->xcur[lane];
    // GetCovariance (:392-433): (J~^T J~)^-1 of the last built problem at the final parameters
    const double a = E.h00, b = E.h01, c = E.h02, d = E.h11, e = E.h12, f = E.h22;
    const double C00 = d * f - e * e, C01 = c * e - b * f, C02 = b * e - c * d;
    const double det = a * C00 + b * C01 + c * C02;
    const int dof = sh->nres - 3;
    const bool ok = have_cov && det > 0 && isfinite(det) && dof != 0;
    ret = ok ? 1 : 0;
    if (io.cov6 && lane < 36) {
      double v = 0;  // entry `lane` of the 6 x 6 matrix
      if (ok) {
        const double sc = 30 * (sh->ss.final_cost / dof) / det;
        v = (lane % 7 == 0) ? 1.0 : 0.0;
        if (lane == 0) v = sc * C00;
        if (lane == 1 || lane == 6) v = sc * C01;
        if (lane == 7) v = sc * (a * f - c * c);
        if (lane == 35) v = sc * (a * d - b * b);
        if (lane == 5 || lane == 30) v = sc * C02;  // (1,5)/(5,1) stay 0 (:426-430)
      } else {
        if (lane == 0 || lane == 7) v = 0.1 * 0.1;  // :173
        if (lane == 35) v = 0.01 * 0.01;
      }
      io.cov6[lane] = v;
    }
  } else if (lane < 3) {
    io.poses[L + lane] = sh->tsrc_last[lane];
  }
  if (io.out && lane < CFEAR_OUTER_LDS && lane < sh->nrec) {  // the per-iteration records kept in LDS (ctl_lm_done)
    const auto& rec = sh->orec[lane];
    io.out->inner_iterations[lane] = rec.inner; io.out->termination[lane] = rec.term; io.out->outer_cost[lane] = rec.cost;
    io.out->outer_pose[lane][0] = rec.pose[0]; io.out->outer_pose[lane][1] = rec.pose[1]; io.out->outer_pose[lane][2] = rec.pose[2];
  }
  if (lane == 0) {
    if (io.out) {
      io.out->success = ret; io.out->usable = sh->success ? 1 : 0; io.out->outer_iterations = sh->itr;
      io.out->num_residuals = sh->nres; io.out->num_residual_blocks = sh->M; io.out->final_cost = sh->ss.final_cost;
      io.out->assoc_path = sh->assoc_path;
      io.out->score = sh->success ? sh->ss.final_cost / sh->nres : 0.0;
    }
    sh->ret = ret;
  }
  sh->cmd = REG_CMD_DONE;
}

// The state functions are leaves (no call inside: a function that calls another one saves and restores a register through
// scratch, a round trip to memory on the controller's serial chain at every exit) and return what has to happen next;
// ctl_step, inlined into the kernel, chains them.
enum { CTL_WAIT = 0, CTL_LM_NEXT, CTL_LM_DONE, CTL_BUILD, CTL_FINISH_E, CTL_FINISH_G, CTL_FINISH_NONE, CTL_EVAL_CAND, CTL_IT0 };

// end of one ceres::Solve: the body of the association loop after SolveOptimizationProblem (:117-151)
//
// Exact repeats are not recomputed. A solve that accepted no step (the first candidate already meets a tolerance: the usual
// end of a registration) leaves the pose bit-identical to the one its problem was built at. The next outer iteration of the
// reference then re-associates at the same pose with the same radius (itr >= 2), gets the same residual blocks, the same normal
// equations, the same candidate and the same convergence decision: its summary equals this one's, field by field. So when
// `moved` is clear the loop below takes the bookkeeping of that next iteration (record, score comparison, break rules,
// itr) from the unchanged solver summary instead of running build + evaluations again; the match array and the normal
// equations at xcur (what the covariance needs) are still those of the last problem built, as they would be. Typical
// registrations end [.., .., 1, 1] inner iterations: the fourth outer iteration costs nothing.
__device__ __noinline__ int ctl_lm_done(LRegShared* sh) {
  const auto& io = sh->rio;
  const auto& P = sh->rp;
  for (;;) {
    const int itr = sh->itr;
    sh->success = (sh->ss.termination != 2);
    if (sh->success) { sh->tsrc_last[0] = sh->xcur[0]; sh->tsrc_last[1] = sh->xcur[1]; sh->tsrc_last[2] = sh->xcur[2]; }
    if (itr - 1 < CFEAR_OUTER_LDS) {  // (every lane stores the same values)
      auto& rec = sh->orec[itr - 1];
      rec.inner = sh->ss.num_iterations; rec.term = sh->ss.termination; rec.cost = sh->ss.final_cost;
      rec.pose[0] = sh->xcur[0]; rec.pose[1] = sh->xcur[1]; rec.pose[2] = sh->xcur[2];
      sh->nrec = itr;
    } else if (lane_id() == 0 && io.out && itr - 1 < CFEAR_MAX_OUTER) {
      io.out->inner_iterations[itr - 1] = sh->ss.num_iterations; io.out->termination[itr - 1] = sh->ss.termination;
      io.out->outer_cost[itr - 1] = sh->ss.final_cost;
      io.out->outer_pose[itr - 1][0] = sh->xcur[0]; io.out->outer_pose[itr - 1][1] = sh->xcur[1]; io.out->outer_pose[itr - 1][2] = sh->xcur[2];
    }
    const double current_score = sh->ss.final_cost;
    const double rel_improvement = (sh->prev_score - current_score) / sh->prev_score;
    bool brk = false, reverted = false;
    if (itr > P.min_itr) {  // :134-149
      if (sh->prev_score < current_score) {
        sh->xcur[0] = sh->prev_par[0]; sh->xcur[1] = sh->prev_par[1]; sh->xcur[2] = sh->prev_par[2]; sh->cur_c = sh->prev_c; sh->cur_s = sh->prev_s;
        brk = true; reverted = true;
      }
      else if (rel_improvement < 0.00001) brk = true;
      else if (sh->ss.last_relative_decrease < 0.00001 || sh->ss.num_iterations == 1) brk = true;
    }
    if (!brk) {
      sh->prev_score = current_score;
      sh->prev_par[0] = sh->xcur[0]; sh->prev_par[1] = sh->xcur[1]; sh->prev_par[2] = sh->xcur[2]; sh->prev_c = sh->cur_c; sh->prev_s = sh->cur_s;
      sh->itr = itr + 1;  // for-loop increment (:102)
      if (sh->itr <= P.max_outer && sh->success) {
        if (!sh->moved && itr >= 2 && !P.recompute_repeats) continue;  // the next outer iteration is an exact repeat of this one (see above)
        return CTL_BUILD;
      }
    }
    // loop left: covariance of the last built problem at the final parameters if the solution is usable (:164-183). The LM
    // state already holds the normal equations of that problem at xcur (every accepted step stores them) unless the
    // parameters were just reverted to the previous outer iteration's: only then is another evaluation needed.
    if (sh->success && reverted) { ctl_publish_eval_cur(sh, REG_ST_COV); return CTL_WAIT; }
    return sh->success ? CTL_FINISH_E : CTL_FINISH_NONE;
  }
}

// trust-region step(s) until a candidate needs evaluating or the solve ends
__device__ __forceinline__ int ctl_lm_next_body(LRegShared* sh) {
  const auto& P = sh->rp;
  const double min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32, min_radius = 1e-32;
  for (;;) {
    if (sh->iteration >= P.max_inner) { sh->ss.termination = 1; return CTL_LM_DONE; }
    if (sh->radius < min_radius) { sh->ss.termination = 0; return CTL_LM_DONE; }
    sh->iteration++;
    const NormalEq E = neq_load(&sh->E);
    const double sc0 = sh->sc0, sc1 = sh->sc1, sc2 = sh->sc2;
    double Hs[6], gs[3];
    Hs[0] = E.h00 * sc0 * sc0; Hs[1] = E.h01 * sc0 * sc1; Hs[2] = E.h02 * sc0 * sc2;
    Hs[3] = E.h11 * sc1 * sc1; Hs[4] = E.h12 * sc1 * sc2; Hs[5] = E.h22 * sc2 * sc2;
    gs[0] = E.g0 * sc0; gs[1] = E.g1 * sc1; gs[2] = E.g2 * sc2;
    if (!sh->reuse_diagonal) {
      sh->dg0 = fmin(fmax(Hs[0], min_lm_diagonal), max_lm_diagonal);
      sh->dg1 = fmin(fmax(Hs[3], min_lm_diagonal), max_lm_diagonal);
      sh->dg2 = fmin(fmax(Hs[5], min_lm_diagonal), max_lm_diagonal);
    }
    // D^T D of the LM diagonal D = sqrt(diag / radius): the square root is squared again, so it is left out
    const double inv_radius = div_well_scaled(1.0, sh->radius);  // radius in [1e-32, 1e16]
    const double Am[6] = {Hs[0] + sh->dg0 * inv_radius, Hs[1], Hs[2], Hs[3] + sh->dg1 * inv_radius, Hs[4], Hs[5] + sh->dg2 * inv_radius};
    const double rhs[3] = {-gs[0], -gs[1], -gs[2]};
    double y[3];
    bool valid = chol3_solve(Am, rhs, y);
    sh->reuse_diagonal = 1;
    double mcc = 0;
    if (valid) {
      const double Hy0 = Hs[0] * y[0] + Hs[1] * y[1] + Hs[2] * y[2];
      const double Hy1 = Hs[1] * y[0] + Hs[3] * y[1] + Hs[4] * y[2];
      const double Hy2 = Hs[2] * y[0] + Hs[4] * y[1] + Hs[5] * y[2];
      mcc = -((y[0] * gs[0] + y[1] * gs[1] + y[2] * gs[2]) + 0.5 * (y[0] * Hy0 + y[1] * Hy1 + y[2] * Hy2));
      if (!(mcc > 0.0)) valid = false;
    }
    if (!valid) {  // HandleInvalidStep
      if (++sh->num_invalid >= 5) { sh->ss.termination = 2; return CTL_LM_DONE; }
      sh->radius = sh->radius / sh->decrease_factor; sh->decrease_factor *= 2.0; sh->reuse_diagonal = 1;
      sh->ss.num_iterations++; sh->ss.last_relative_decrease = 0.0;
      if (sh->x_cost < sh->ss.final_cost) sh->ss.final_cost = sh->x_cost;
      continue;
    }
    sh->num_invalid = 0;
    sh->model_cost_change = mcc;
    sh->xc[0] = sh->xcur[0] + y[0] * sc0; sh->xc[1] = sh->xcur[1] + y[1] * sc1; sh->xc[2] = sh->xcur[2] + y[2] * sc2;
    return CTL_EVAL_CAND;  // the candidate is published by a function of its own: sincos on top of this one's registers reaches the
                           // callee-saved ones, and saving those is a round trip through scratch at the exit
  }
}

// mahalanobisDistanceError (n_scan_normal.h:259-290) at x: r = L (alpha (guess - x)), J = -alpha L, no loss
// (forceinline, by value: a NormalEq handed to an out-of-line function by reference would live in per-thread scratch)
__device__ __forceinline__ NormalEq add_prior(const LRegShared* sh, NormalEq E, double x0, double x1, double x2) {
  const double a = sh->palpha;
  const double d0 = a * (sh->pguess[0] - x0), d1 = a * (sh->pguess[1] - x1), d2 = a * (sh->pguess[2] - x2);
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const double l0 = sh->pL[3 * i], l1 = sh->pL[3 * i + 1], l2 = sh->pL[3 * i + 2];
    const double r = l0 * d0 + l1 * d1 + l2 * d2;
    const double j0 = -a * l0, j1 = -a * l1, j2 = -a * l2;
    E.cost += 0.5 * (r * r);
    E.g0 += j0 * r; E.g1 += j1 * r; E.g2 += j2 * r;
    E.h00 += j0 * j0; E.h01 += j0 * j1; E.h02 += j0 * j2; E.h11 += j1 * j1; E.h12 += j1 * j2; E.h22 += j2 * j2;
  }
  return E;
}

// ---- one function per controller state (kept out of line: the kernel's register budget is the maximum
// over its callees, and it decides how many workgroups share a compute unit) ----
__device__ __noinline__ int ctl_after_build(LRegShared* sh) {
  const auto& P = sh->rp;
  const int rpb = (P.cost == CFEAR_COST_P2L) ? 1 : 2;
  sh->nres = sh->M * rpb;
  if (sh->nres <= 1) {  // :370-371 -> :114-115
    sh->success = 0;
    return CTL_FINISH_NONE;
  }
  if (sh->prior_on) sh->nres += 3;  // the prior block joins after the residual-count check (:370-377)
  return CTL_IT0;  // the waves evaluated the new problem at x right after building it (register_block): no command of its own
}

__device__ __forceinline__ int ctl_after_it0_body(LRegShared* sh) {
  const double gradient_tolerance = 1e-10;
  NormalEq E = gather_partials_regs(sh->rw.red);  // (in registers: stored once, as the current point's equations)
  if (sh->prior_on) E = add_prior(sh, E, sh->x[0], sh->x[1], sh->x[2]);
  neq_store(&sh->E, E); sh->x_cost = E.cost;
  sh->x_norm = sqrt_well_scaled(sh->xcur[0] * sh->xcur[0] + sh->xcur[1] * sh->xcur[1] + sh->xcur[2] * sh->xcur[2]);
  sh->ss.num_iterations = 1; sh->ss.final_cost = E.cost; sh->ss.last_relative_decrease = 0.0; sh->ss.termination = 1;
  sh->moved = 0;
  const double gmax = fmax(fabs(E.g0), fmax(fabs(E.g1), fabs(E.g2)));
  if (gmax <= gradient_tolerance) { sh->ss.termination = 0; return CTL_LM_DONE; }
  sh->sc0 = 1.0 / (1.0 + sqrt(E.h00)); sh->sc1 = 1.0 / (1.0 + sqrt(E.h11)); sh->sc2 = 1.0 / (1.0 + sqrt(E.h22));
  sh->radius = 1e4; sh->decrease_factor = 2.0; sh->reuse_diagonal = 0; sh->num_invalid = 0; sh->iteration = 0;
  sh->dg0 = sh->dg1 = sh->dg2 = 0;
  return CTL_LM_NEXT;
}

__device__ __forceinline__ int ctl_after_candidate_body(LRegShared* sh) {
  const auto& P = sh->rp;
  const double min_relative_decrease = 1e-3, function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double max_radius = 1e16;
  NormalEq C = gather_partials_regs(sh->rw.red);
  if (sh->prior_on) C = add_prior(sh, C, sh->x[0], sh->x[1], sh->x[2]);
  const double cand_cost = C.cost;
  const double d0 = sh->xcur[0] - sh->xc[0], d1 = sh->xcur[1] - sh->xc[1], d2 = sh->xcur[2] - sh->xc[2];
  const double step_norm = sqrt_well_scaled(d0 * d0 + d1 * d1 + d2 * d2);
  if (step_norm <= parameter_tolerance * (sh->x_norm + parameter_tolerance)) { sh->ss.termination = 0; return CTL_LM_DONE; }
  const double cost_change = sh->x_cost - cand_cost;
  if (fabs(cost_change) <= function_tolerance * sh->x_cost) { sh->ss.termination = 0; return CTL_LM_DONE; }
  const double mcc = sh->model_cost_change;  // > 0 (ctl_lm_next); the short quotient unless it is next to the denormals
  const double relative_decrease = mcc > 1e-290 ? div_well_scaled(cost_change, mcc) : cost_change / mcc;
  sh->ss.num_iterations++;
  sh->ss.last_relative_decrease = relative_decrease;
  if (relative_decrease > min_relative_decrease) {  // HandleSuccessfulStep
    sh->xcur[0] = sh->xc[0]; sh->xcur[1] = sh->xc[1]; sh->xcur[2] = sh->xc[2]; sh->cur_c = sh->c; sh->cur_s = sh->s;
    sh->moved = 1;
    sh->x_norm = sqrt_well_scaled(sh->xcur[0] * sh->xcur[0] + sh->xcur[1] * sh->xcur[1] + sh->xcur[2] * sh->xcur[2]);
    neq_store(&sh->E, C); sh->x_cost = cand_cost;
    const double t = 2.0 * relative_decrease - 1.0;
    sh->radius = div_well_scaled(sh->radius, fmax(1.0 / 3.0, 1.0 - t * t * t));  // divisor in [1/3, 1]
    sh->radius = fmin(max_radius, sh->radius);
    sh->decrease_factor = 2.0; sh->reuse_diagonal = 0;
    if (sh->x_cost < sh->ss.final_cost) sh->ss.final_cost = sh->x_cost;
    const double gmax = fmax(fabs(C.g0), fmax(fabs(C.g1), fabs(C.g2)));
    if (sh->iteration >= P.max_inner) { sh->ss.termination = 1; return CTL_LM_DONE; }
    if (gmax <= gradient_tolerance) { sh->ss.termination = 0; return CTL_LM_DONE; }
  } else {  // HandleUnsuccessfulStep
    sh->radius = sh->radius / sh->decrease_factor; sh->decrease_factor *= 2.0; sh->reuse_diagonal = 1;
    if (cand_cost < sh->ss.final_cost) sh->ss.final_cost = cand_cost;
  }
  return CTL_LM_NEXT;
}

// The state function and the trust-region step that follows it as ONE out-of-line function: what the first leaves in LDS
// (normal equations, radius, scaling) the second finds in registers - no store -> load round trips, one call less.
__device__ __noinline__ int ctl_lm_next(LRegShared* sh) { return ctl_lm_next_body(sh); }
__device__ __noinline__ int ctl_after_it0(LRegShared* sh) {
  const int nx = ctl_after_it0_body(sh);
  return nx == CTL_LM_NEXT ? ctl_lm_next_body(sh) : nx;
}
__device__ __noinline__ int ctl_after_candidate(LRegShared* sh) {
  const int nx = ctl_after_candidate_body(sh);
  return nx == CTL_LM_NEXT ? ctl_lm_next_body(sh) : nx;
}

__device__ __noinline__ int ctl_after_cov(LRegShared* sh) {
  gather_partials(sh->rw.red, &sh->G);
  if (sh->prior_on) neq_store(&sh->G, add_prior(sh, neq_load(&sh->G), sh->x[0], sh->x[1], sh->x[2]));
  return CTL_FINISH_G;
}

// consumes the result of the command just executed and publishes the next one. acc (tools, timed instantiation only):
// accumulators [4..7] += time in the state function, in ctl_lm_next, in the publishing function, in ctl_lm_done
__device__ __forceinline__ void ctl_step(LRegShared* sh, long long* acc = nullptr) {  // acc: registers of the caller (static indices)
  int nx;
  long long t0 = 0;
  if (acc) t0 = (long long)wall_clock64();
  switch (sh->state) {
    case REG_ST_BUILD: nx = ctl_after_build(sh); if (nx == CTL_IT0) nx = ctl_after_it0(sh); break;
    case REG_ST_LM_IT0: nx = ctl_after_it0(sh); break;
    case REG_ST_LM_CAND: nx = ctl_after_candidate(sh); break;
    default: nx = ctl_after_cov(sh); break;
  }
  if (acc) { const long long t = (long long)wall_clock64(); acc[4] += t - t0; t0 = t; }
  while (nx != CTL_WAIT) {
    int slot = 6;
    switch (nx) {
      case CTL_LM_NEXT: nx = ctl_lm_next(sh); slot = 5; break;
      case CTL_LM_DONE: nx = ctl_lm_done(sh); slot = 7; break;
      case CTL_BUILD: ctl_publish_build(sh, false); nx = CTL_WAIT; break;
      case CTL_EVAL_CAND: ctl_publish_candidate(sh); nx = CTL_WAIT; break;
      case CTL_FINISH_E: ctl_finish(sh, true, &sh->E); nx = CTL_WAIT; break;
      case CTL_FINISH_G: ctl_finish(sh, true, &sh->G); nx = CTL_WAIT; break;
      default: ctl_finish(sh, false, &sh->E); nx = CTL_WAIT; break;
    }
    if (acc) {
      const long long t = (long long)wall_clock64(), d = t - t0;
      acc[5] += slot == 5 ? d : 0; acc[6] += slot == 6 ? d : 0; acc[7] += slot == 7 ? d : 0;
      t0 = t;
    }
  }
}

// n_scan_normal_reg::Register. poses: n x 3 in global memory (in/out); cov6: 36 doubles or null;
// out: summary in global memory. par_lds: >= 3*n doubles of LDS; sh: RegShared in LDS.
// Wave 0 is the controller; every wave executes the published commands (two barriers per command).
template <int KCOST = -1>
__device__ inline int register_block(ScanDev* const* scans, int n, double* poses, double* cov6, const RegParams& P_in,
                                     const RegScratch& W_in, double* par_lds, RegShared* sh, cfear_reg_summary* out,
                                     PhaseTimer* pt = nullptr, const double* prior_cov6 = nullptr) {
  const int tid = threadIdx.x;
  LRegShared* ls = (LRegShared*)sh;  // the same object through an LDS-typed pointer (see LRegShared)
  if (tid == 0) {
    sh->rp = P_in; sh->rw = W_in;
    sh->rio.poses = poses; sh->rio.cov6 = cov6; sh->rio.out = out; sh->rio.par = par_lds; sh->rio.n = n;
  }
  __syncthreads();
  const RegParams& P = CFEAR_GENERIC(const RegParams, sh->rp);
  const RegScratch& W = CFEAR_GENERIC(const RegScratch, sh->rw);
  const RegIo& io = sh->rio;
  const bool master = (tid >> 6) == 0;
  // the last pose as passed in (what a failed registration hands back): read before the normalisation below, which works in
  // place when the caller keeps its poses in par_lds (the step kernel does: no trip through memory for them)
  // (read by the controller wave, which also holds the thread that normalises that entry - n <= 64 - so program order is enough)
  double last_in0 = 0, last_in1 = 0, last_in2 = 0;
  if (master) { last_in0 = poses[3 * (n - 1)]; last_in1 = poses[3 * (n - 1) + 1]; last_in2 = poses[3 * (n - 1) + 2]; }
  // Affine3dToVectorXYeZ(Tsrc[i]) (:88-92): theta -> atan2(sin, cos)
  for (int i = tid; i < n; i += CFEAR_REG_BLOCK) {
    const Aff2 T = aff_from_xyt(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]);
    double v[3]; aff_to_xyt(T, v);
    par_lds[3 * i] = v[0]; par_lds[3 * i + 1] = v[1]; par_lds[3 * i + 2] = v[2];
  }
  for (int i = tid; i < n; i += CFEAR_REG_BLOCK) sh->kf[i] = grid_view(scans[i]);
  if (tid == 64) { sh->srs = scans[n - 1]->rsrc; sh->scc = (long long)scans[n - 1]->cap_cells; }
  if (tid == 0 && out) {
    out->success = 0; out->usable = 0; out->outer_iterations = 0; out->num_residuals = 0; out->num_residual_blocks = 0;
    out->assoc_path = 0; out->final_cost = 0; out->score = 0;
  }
  if (out)  // one thread per outer iteration (448 stores by a single thread were a measurable part of the start-up)
    for (int i = tid; i < CFEAR_MAX_OUTER; i += CFEAR_REG_BLOCK) { out->inner_iterations[i] = 0; out->termination[i] = 0; out->outer_cost[i] = 0; out->outer_pose[i][0] = out->outer_pose[i][1] = out->outer_pose[i][2] = 0; }
  __syncthreads();
  if (master) {
    const int L = 3 * (n - 1);
    sh->xcur[0] = par_lds[L]; sh->xcur[1] = par_lds[L + 1]; sh->xcur[2] = par_lds[L + 2];
    sh->prior_on = 0;
    if (prior_cov6) {  // :373-376: guess_inf_sqrt = Cov6to3(cov).inverse().llt().matrixL(), alpha = sqrt(#source cells)
      const double* Cq = prior_cov6;
      const double a = Cq[0], b = Cq[1], c = Cq[5], d = Cq[6], e = Cq[7], f5 = Cq[11], g6 = Cq[30], h = Cq[31], i9 = Cq[35];  // registration.cpp:123-129
      const double A00 = e * i9 - f5 * h, A10 = f5 * g6 - d * i9, A11 = a * i9 - c * g6, A20 = d * h - e * g6, A21 = b * g6 - a * h, A22 = a * e - b * d;
      const double det = a * A00 + b * A10 + c * A20;
      const double i00 = A00 / det, i10 = A10 / det, i11 = A11 / det, i20 = A20 / det, i21 = A21 / det, i22 = A22 / det;  // lower triangle of the inverse
      const double l00 = sqrt(i00), l10 = i10 / l00, l20 = i20 / l00;
      const double l11 = sqrt(i11 - l10 * l10), l21 = (i21 - l20 * l10) / l11;
      const double l22 = sqrt(i22 - l20 * l20 - l21 * l21);
      sh->pL[0] = l00; sh->pL[1] = 0; sh->pL[2] = 0; sh->pL[3] = l10; sh->pL[4] = l11; sh->pL[5] = 0; sh->pL[6] = l20; sh->pL[7] = l21; sh->pL[8] = l22;
      sh->pguess[0] = par_lds[L]; sh->pguess[1] = par_lds[L + 1]; sh->pguess[2] = par_lds[L + 2];  // Affine3dToEigVectorXYeZ(Tsrc.back()) (:93-94)
      sh->palpha = sqrt((double)scans[n - 1]->n_cells);
      sh->prior_on = 1;
    }
    sh->prev_par[0] = sh->xcur[0]; sh->prev_par[1] = sh->xcur[1]; sh->prev_par[2] = sh->xcur[2];
    sh->tsrc_last[0] = last_in0; sh->tsrc_last[1] = last_in1; sh->tsrc_last[2] = last_in2;
    sh->prev_score = 1.7976931348623157e308;
    sh->success = 1; sh->nres = 0; sh->M = 0; sh->ret = 0; sh->itr = 1; sh->nrec = 0;
    sh->moved = 1;  // (set by every solve's first evaluation, ctl_after_it0_body; a conservative value until then)
    sh->ss.num_iterations = 0; sh->ss.termination = 0; sh->ss.final_cost = 0; sh->ss.last_relative_decrease = 0;
    ctl_publish_build(ls, true);
  }
  // tools (timed instantiation, phase_detail 2): where a command's time goes, seen by thread 0. acc2[0..3] += wait at the
  // command barrier, execution of the command by wave 0, wait at the result barrier, commands; [4..7]: see ctl_step
  long long* const acc2g = (pt && pt->acc2 && tid == 0) ? pt->acc2 : nullptr;
  long long ac[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // (in registers; stored once at the end)
  long long* const acc2 = acc2g ? ac : nullptr;
  for (;;) {
    long long t0c = 0;
    if (pt && (pt->acc || acc2) && tid == 0) t0c = (long long)wall_clock64();
    __syncthreads();  // command visible to every wave
    const int cmd = ls->cmd;
    if (cmd == REG_CMD_DONE) break;
    long long tb = 0;
    if (acc2) tb = (long long)wall_clock64();
    if (cmd == REG_CMD_BUILD) {
      if (pt) pt->mark();
      const int M = build_problem_block<KCOST>(scans, n, ls, ls->itr);
      if (tid == 0) ls->M = M;
      // the first evaluation of the solve that follows, at the pose the problem was built for: straight away instead of as a
      // command of its own (a barrier pair and a turn of the controller less per outer iteration); block-uniform condition,
      // the same as ctl_after_build's
      if (M * (((KCOST >= 0 ? KCOST : ls->rp.cost) == CFEAR_COST_P2L) ? 1 : 2) > 1) evaluate_partial<KCOST>(ls, M, ls->lds_match, ls->x[0], ls->x[1], ls->c, ls->s);
      if (pt) pt->mark();
    } else {
      evaluate_partial<KCOST>(ls, ls->M, ls->lds_match, ls->x[0], ls->x[1], ls->c, ls->s);
    }
    long long t1 = 0;
    if (pt && (pt->acc || acc2) && tid == 0) t1 = (long long)wall_clock64();
    __syncthreads();  // results visible to the controller
    if (acc2 && cmd != REG_CMD_BUILD) { const long long t = (long long)wall_clock64(); acc2[0] += tb - t0c; acc2[1] += t1 - tb; acc2[2] += t - t1; acc2[3] += 1; }
    if (master) {
      __builtin_amdgcn_s_setprio(3);  // the serial chain of the workgroup: ahead of the other workgroups' waves on this SIMD
      ctl_step(ls, (cmd != REG_CMD_BUILD) ? acc2 : nullptr);
      __builtin_amdgcn_s_setprio(0);
    }
    if (pt && pt->acc && tid == 0 && cmd != REG_CMD_BUILD) {  // tools: time in evaluations (incl. the barrier before) and in the controller
      const long long t2 = (long long)wall_clock64();
      pt->acc[0] += t1 - t0c; pt->acc[1] += t2 - t1; pt->acc[2] += 1;
    }
  }
  if (acc2g) for (int i = 0; i < 8; i++) acc2g[i] = ac[i];
  const int ret = ls->ret;
  __syncthreads();
  return ret;
}

// n_scan_normal_reg::GetCost (n_scan_normal.cpp:188-213): associations and residual blocks at the given poses
// (BuildOptimizationProblem with the caller's itr_, which only selects the association radius, :222), then
// ceres::Problem::Evaluate with default options: *score = 1/2 sum rho, residuals = sqrt(rho') r per residual in
// residual-block order. *n_res = number of residuals, or -1 where the reference returns false (<= 1 residuals).
__device__ inline void get_cost_block(ScanDev* const* scans, int n, const double* poses, const RegParams& P_in, const RegScratch& W_in,
                                      double* par_lds, RegShared* sh, int itr, double* score, double* residuals, int cap, int* n_res) {
  const int tid = threadIdx.x;
  LRegShared* ls = (LRegShared*)sh;
  if (tid == 0) {
    sh->rp = P_in; sh->rw = W_in;
    sh->rio.poses = nullptr; sh->rio.cov6 = nullptr; sh->rio.out = nullptr; sh->rio.par = par_lds; sh->rio.n = n;
  }
  for (int i = tid; i < n; i += CFEAR_REG_BLOCK) {  // Affine3dToVectorXYeZ (:196)
    const Aff2 T = aff_from_xyt(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]);
    double v[3]; aff_to_xyt(T, v);
    par_lds[3 * i] = v[0]; par_lds[3 * i + 1] = v[1]; par_lds[3 * i + 2] = v[2];
    sh->kf[i] = grid_view(scans[i]);
  }
  if (tid == 64) { sh->srs = scans[n - 1]->rsrc; sh->scc = (long long)scans[n - 1]->cap_cells; }
  __syncthreads();
  const RegParams& P = CFEAR_GENERIC(const RegParams, sh->rp);
  const RegScratch& W = CFEAR_GENERIC(const RegScratch, sh->rw);
  if ((tid >> 6) == 0) {
    const int L = 3 * (n - 1);
    sh->xcur[0] = par_lds[L]; sh->xcur[1] = par_lds[L + 1]; sh->xcur[2] = par_lds[L + 2];
    sh->prior_on = 0;
    ctl_publish_build(ls, true);
  }
  __syncthreads();
  const int M = build_problem_block(scans, n, ls, itr);
  const int nres = M * ((P.cost == CFEAR_COST_P2L) ? 1 : 2);
  if (nres <= 1) {  // :205-208
    if (tid == 0) { *n_res = -1; *score = 0.0; }
    return;
  }
  double sn, cs;
  sincos(sh->xcur[2], &sn, &cs);
  evaluate_partial(ls, M, ls->lds_match, ls->xcur[0], ls->xcur[1], cs, sn, residuals, cap);
  __syncthreads();
  if (tid == 0) {
    gather_partials(ls->rw.red, &ls->G);
    *score = ls->G.cost; *n_res = nres;
  }
}

}  // namespace cfear_dev
