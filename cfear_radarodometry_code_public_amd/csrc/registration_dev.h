// registration_dev.h -- K6/K7/K8 device code: scan-to-keyframes registration of one problem by one
// workgroup, entirely on the device (association, robust normal equations, Levenberg-Marquardt
// controller, outer re-association loop, covariance). Replaces n_scan_normal_reg::Register
// (n_scan_normal.cpp:82-187) + AddScanPairCost (:215-326) + ceres::Solve (:443-452) + GetCovariance
// (:392-433) for this path. Ceres trust-region LM semantics restated per SURVEY.md 9.H; same
// formulas, in double, as oracle/cfear_oracle.c so that iteration counts agree.
//
// All threads of the block keep identical copies of the solver state (uniform control flow); the
// only data-parallel parts are the association pass and the residual pass, each followed by a
// deterministic block reduction.
#pragma once
#include "features_dev.h"

namespace cfear_dev {

struct RegParams {
  int cost, loss, weight_opt;
  double loss_limit, covar_scale, regularization, assoc_radius;
  int max_outer, min_itr, max_inner;
};

// Per-block global scratch: compacted matches (SoA) + per-pair association result.
struct RegScratch {
  double* tmx; double* tmy;  // Ttar * tar_mean
  double* a0; double* a1; double* a2;  // P2L: (n_x, n_y, -) ; P2D: (l00, l10, l11)
  double* sx; double* sy;    // source mean (local frame)
  double* w;                 // weight after loss
  int* assoc;                // [pairs] target cell index or -1
  float* sim;                // [pairs] direction similarity
  int cap;                   // capacity in pairs
  double* red;               // LDS, >= 10 * 32 doubles
  int* red_i;                // LDS, >= 64 ints
};

struct Aff2 { double l0, l1, l2, l3, t0, t1; };

__device__ inline Aff2 aff_from_xyt(double x, double y, double th) {  // vectorToAffine3d, registration.cpp:130-136
  Aff2 T; const double c = cos(th), s = sin(th);
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

__device__ inline double similarity(double x, double y) { return 2 * fmin(x, y) / (x + y); }  // registration.h:96
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
__device__ inline void loss_eval(int loss, double a, double s, double rho[3]) {
  const double b = a * a;
  switch (loss) {
    case CFEAR_LOSS_HUBER:
      if (s > b) { const double r = sqrt(s); rho[0] = 2.0 * a * r - b; rho[1] = fmax(CFEAR_DBL_MIN, a / r); rho[2] = -rho[1] / (2.0 * s); }
      else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
      return;
    case CFEAR_LOSS_CAUCHY: {
      const double c = 1.0 / b, sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = b * log(sum); rho[1] = fmax(CFEAR_DBL_MIN, inv); rho[2] = -c * (inv * inv);
      return; }
    case CFEAR_LOSS_SOFTLONE: {
      const double c = 1.0 / b, sum = 1.0 + s * c, tmp = sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0); rho[1] = fmax(CFEAR_DBL_MIN, 1.0 / tmp); rho[2] = -(c * rho[1]) / (2.0 * sum);
      return; }
    case CFEAR_LOSS_TUKEY:
      if (s <= b) { const double v = 1.0 - s / b, v2 = v * v; rho[0] = b / 3.0 * (1.0 - v2 * v); rho[1] = v2; rho[2] = -2.0 / b * v; }
      else { rho[0] = b / 3.0; rho[1] = 0.0; rho[2] = 0.0; }
      return;
    case CFEAR_LOSS_COMBINED: {
      double g[3], f[3];
      { const double sum = 1.0 + s, inv = 1.0 / sum; g[0] = log(sum); g[1] = fmax(CFEAR_DBL_MIN, inv); g[2] = -(inv * inv); }
      if (g[0] > 1.0) { const double r = sqrt(g[0]); f[0] = 2.0 * r - 1.0; f[1] = fmax(CFEAR_DBL_MIN, 1.0 / r); f[2] = -f[1] / (2.0 * g[0]); }
      else { f[0] = g[0]; f[1] = 1.0; f[2] = 0.0; }
      rho[0] = f[0]; rho[1] = f[1] * g[1]; rho[2] = f[2] * g[1] * g[1] + f[1] * g[2];
      return; }
    default: rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return;
  }
}

struct NormalEq { double cost, g0, g1, g2, h00, h01, h02, h11, h12, h22; };

// block reduction of the 10 accumulators; result identical in every thread
__device__ inline NormalEq block_reduce_neq(NormalEq v, double* red) {
  double a[10] = {v.cost, v.g0, v.g1, v.g2, v.h00, v.h01, v.h02, v.h11, v.h12, v.h22};
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
    for (int i = 0; i < 10; i++) a[i] += __shfl_xor(a[i], off);
  }
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane_id() == 0) {
#pragma unroll
    for (int i = 0; i < 10; i++) red[i * 32 + w] = a[i];
  }
  __syncthreads();
  double r[10];
#pragma unroll
  for (int i = 0; i < 10; i++) {
    double s = 0;
    for (int j = 0; j < nw; j++) s += red[i * 32 + j];
    r[i] = s;
  }
  NormalEq o;
  o.cost = r[0]; o.g0 = r[1]; o.g1 = r[2]; o.g2 = r[3]; o.h00 = r[4]; o.h01 = r[5]; o.h02 = r[6]; o.h11 = r[7]; o.h12 = r[8]; o.h22 = r[9];
  return o;
}

// Robustified cost, gradient and Gauss-Newton matrix over the compacted matches at x = (x, y, theta).
// Residuals: n_scan_normal.h:190-201 (P2L), :224-243 (P2D), :336-350 (P2P); corrector = sqrt(rho').
__device__ inline NormalEq evaluate_block(const RegScratch& W, int M, const RegParams& P, double x0, double x1, double x2) {
  const double c = cos(x2), s = sin(x2);
  NormalEq a = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < M; i += blockDim.x) {
    const double sx = W.sx[i], sy = W.sy[i], tmx = W.tmx[i], tmy = W.tmy[i], wgt = W.w[i];
    const double px = (c * sx - s * sy) + x0;
    const double py = (s * sx + c * sy) + x1;
    const double dtx = -s * sx - c * sy;
    const double dty = c * sx - s * sy;
    double r[2], J[2][3];
    int nr;
    if (P.cost == CFEAR_COST_P2L) {
      const double nx = W.a0[i], ny = W.a1[i];
      nr = 1;
      r[0] = (px - tmx) * nx + (py - tmy) * ny;
      J[0][0] = nx; J[0][1] = ny; J[0][2] = dtx * nx + dty * ny;
      r[1] = 0; J[1][0] = J[1][1] = J[1][2] = 0;
    } else if (P.cost == CFEAR_COST_P2D) {
      const double l00 = W.a0[i], l10 = W.a1[i], l11 = W.a2[i];
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
    double rho[3];
    loss_eval(P.loss, P.loss_limit, sq, rho);
    rho[0] *= wgt; rho[1] *= wgt;  // ScaledLoss (n_scan_normal.cpp:277)
    a.cost += 0.5 * rho[0];
    const double sr = sqrt(rho[1]);
    for (int k = 0; k < nr; k++) {
      const double rk = sr * r[k];
      const double j0 = sr * J[k][0], j1 = sr * J[k][1], j2 = sr * J[k][2];
      a.g0 += j0 * rk; a.g1 += j1 * rk; a.g2 += j2 * rk;
      a.h00 += j0 * j0; a.h01 += j0 * j1; a.h02 += j0 * j2;
      a.h11 += j1 * j1; a.h12 += j1 * j2; a.h22 += j2 * j2;
    }
  }
  return block_reduce_neq(a, W.red);
}

__device__ inline bool chol3_solve(const double A[6], const double b[3], double y[3]) {
  const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
  if (!(a00 > 0)) return false;
  const double l00 = sqrt(a00), l10 = a01 / l00, l20 = a02 / l00;
  const double d1 = a11 - l10 * l10;
  if (!(d1 > 0)) return false;
  const double l11 = sqrt(d1), l21 = (a12 - l20 * l10) / l11;
  const double d2 = a22 - l20 * l20 - l21 * l21;
  if (!(d2 > 0)) return false;
  const double l22 = sqrt(d2);
  const double z0 = b[0] / l00, z1 = (b[1] - l10 * z0) / l11, z2 = (b[2] - l20 * z0 - l21 * z1) / l22;
  y[2] = z2 / l22; y[1] = (z1 - l21 * y[2]) / l11; y[0] = (z0 - l10 * y[1] - l20 * y[2]) / l00;
  return isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2]);
}

struct SolveSummary { int num_iterations; int termination; double final_cost; double last_relative_decrease; };

// ceres::Solve restatement: trust-region LM, Jacobi scaling, default tolerances, max_inner iterations.
// One residual pass per LM iteration: cost, gradient and JtJ are evaluated together at the candidate
// point (Ceres evaluates the cost first and the Jacobian after acceptance: same values).
__device__ inline SolveSummary lm_solve_block(const RegScratch& W, int M, const RegParams& P, double x[3]) {
  const double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double max_radius = 1e16, min_radius = 1e-32;
  SolveSummary S;
  NormalEq E = evaluate_block(W, M, P, x[0], x[1], x[2]);
  double x_cost = E.cost;
  double x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  S.num_iterations = 1; S.final_cost = x_cost; S.last_relative_decrease = 0.0; S.termination = 1;
  double gmax = fmax(fabs(E.g0), fmax(fabs(E.g1), fabs(E.g2)));
  if (gmax <= gradient_tolerance) { S.termination = 0; return S; }
  const double sc0 = 1.0 / (1.0 + sqrt(E.h00)), sc1 = 1.0 / (1.0 + sqrt(E.h11)), sc2 = 1.0 / (1.0 + sqrt(E.h22));
  double radius = 1e4, decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int num_invalid = 0, iteration = 0;
  double dg0 = 0, dg1 = 0, dg2 = 0;
  for (;;) {
    if (iteration >= P.max_inner) { S.termination = 1; return S; }
    if (radius < min_radius) { S.termination = 0; return S; }
    iteration++;
    double Hs[6], gs[3];
    Hs[0] = E.h00 * sc0 * sc0; Hs[1] = E.h01 * sc0 * sc1; Hs[2] = E.h02 * sc0 * sc2;
    Hs[3] = E.h11 * sc1 * sc1; Hs[4] = E.h12 * sc1 * sc2; Hs[5] = E.h22 * sc2 * sc2;
    gs[0] = E.g0 * sc0; gs[1] = E.g1 * sc1; gs[2] = E.g2 * sc2;
    if (!reuse_diagonal) {
      dg0 = fmin(fmax(Hs[0], min_lm_diagonal), max_lm_diagonal);
      dg1 = fmin(fmax(Hs[3], min_lm_diagonal), max_lm_diagonal);
      dg2 = fmin(fmax(Hs[5], min_lm_diagonal), max_lm_diagonal);
    }
    const double lm0 = sqrt(dg0 / radius), lm1 = sqrt(dg1 / radius), lm2 = sqrt(dg2 / radius);
    const double Am[6] = {Hs[0] + lm0 * lm0, Hs[1], Hs[2], Hs[3] + lm1 * lm1, Hs[4], Hs[5] + lm2 * lm2};
    const double rhs[3] = {-gs[0], -gs[1], -gs[2]};
    double y[3];
    bool valid = chol3_solve(Am, rhs, y);
    reuse_diagonal = true;
    double model_cost_change = 0;
    if (valid) {
      const double Hy0 = Hs[0] * y[0] + Hs[1] * y[1] + Hs[2] * y[2];
      const double Hy1 = Hs[1] * y[0] + Hs[3] * y[1] + Hs[4] * y[2];
      const double Hy2 = Hs[2] * y[0] + Hs[4] * y[1] + Hs[5] * y[2];
      model_cost_change = -((y[0] * gs[0] + y[1] * gs[1] + y[2] * gs[2]) + 0.5 * (y[0] * Hy0 + y[1] * Hy1 + y[2] * Hy2));
      if (!(model_cost_change > 0.0)) valid = false;
    }
    if (!valid) {
      if (++num_invalid >= 5) { S.termination = 2; return S; }
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      S.num_iterations++; S.last_relative_decrease = 0.0;
      if (x_cost < S.final_cost) S.final_cost = x_cost;
      continue;
    }
    num_invalid = 0;
    const double xc0 = x[0] + y[0] * sc0, xc1 = x[1] + y[1] * sc1, xc2 = x[2] + y[2] * sc2;
    const NormalEq C = evaluate_block(W, M, P, xc0, xc1, xc2);
    const double cand_cost = C.cost;
    const double d0 = x[0] - xc0, d1 = x[1] - xc1, d2 = x[2] - xc2;
    const double step_norm = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { S.termination = 0; return S; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= function_tolerance * x_cost) { S.termination = 0; return S; }
    const double relative_decrease = cost_change / model_cost_change;
    S.num_iterations++;
    S.last_relative_decrease = relative_decrease;
    if (relative_decrease > min_relative_decrease) {
      x[0] = xc0; x[1] = xc1; x[2] = xc2;
      x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      E = C; x_cost = cand_cost;
      radius = radius / fmax(1.0 / 3.0, 1.0 - pow(2.0 * relative_decrease - 1.0, 3.0));
      radius = fmin(max_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = false;
      if (x_cost < S.final_cost) S.final_cost = x_cost;
      gmax = fmax(fabs(E.g0), fmax(fabs(E.g1), fabs(E.g2)));
      if (iteration >= P.max_inner) { S.termination = 1; return S; }
      if (gmax <= gradient_tolerance) { S.termination = 0; return S; }
    } else {
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = true;
      if (cand_cost < S.final_cost) S.final_cost = cand_cost;
    }
  }
}

// AddScanPairCost for every (keyframe i -> current) pair (n_scan_normal.cpp:215-326, :359-367).
// par = n x 3 poses in LDS/global (read only here). Returns the number of matches (compacted in W).
__device__ inline int build_problem_block(ScanDev* const* scans, int n, const double* par, const RegParams& P, int itr,
                                          const RegScratch& W) {
  const double angle_outlier = 0.86602540378443864676;  // cos(M_PI/6)
  const double curr_radius = (itr == 1) ? 2 * P.assoc_radius : P.assoc_radius;  // :222
  const ScanDev* src = scans[n - 1];
  const int nsrc = src->n_cells;
  const int pairs = (n - 1) * nsrc;
  const Aff2 Tsrc = aff_from_xyt(par[3 * (n - 1)], par[3 * (n - 1) + 1], par[3 * (n - 1) + 2]);
  const int ipt = (pairs + blockDim.x - 1) / blockDim.x;
  const int p0 = threadIdx.x * ipt, p1 = min(pairs, p0 + ipt);
  int cnt = 0;
  int cur_i = -1;
  Aff2 T = aff_identity();
  const ScanDev* tar = nullptr;
  for (int p = p0; p < p1; p++) {
    const int i = p / nsrc, j = p - i * nsrc;
    if (i != cur_i) {
      cur_i = i; tar = scans[i];
      const Aff2 Ttar = aff_from_xyt(par[3 * i], par[3 * i + 1], par[3 * i + 2]);
      T = aff_mul(aff_inv(Ttar), Tsrc);  // Tsrctotar (:224)
    }
    const cfear_cell* cs = &src->cells[j];
    const double qx = (T.l0 * cs->mean[0] + T.l1 * cs->mean[1]) + T.t0;
    const double qy = (T.l2 * cs->mean[0] + T.l3 * cs->mean[1]) + T.t1;
    int ti = scan_closest(tar, qx, qy, curr_radius);
    float simf = 0.f;
    if (ti >= 0) {
      const cfear_cell* ct = &tar->cells[ti];
      const double nx = T.l0 * cs->normal[0] + T.l1 * cs->normal[1];
      const double ny = T.l2 * cs->normal[0] + T.l3 * cs->normal[1];
      const double sim = fmax(nx * ct->normal[0] + ny * ct->normal[1], 0.0);
      if (!(sim > angle_outlier)) ti = -1;  // :247
    }
    W.assoc[p] = ti;
    cnt += (ti >= 0) ? 1 : 0;
  }
  int M;
  int o = block_exclusive_scan(cnt, W.red_i, &M);
  cur_i = -1;
  Aff2 Ttar = aff_identity();
  for (int p = p0; p < p1; p++) {
    const int ti = W.assoc[p];
    if (ti < 0) continue;
    const int i = p / nsrc, j = p - i * nsrc;
    if (i != cur_i) {
      cur_i = i; tar = scans[i];
      Ttar = aff_from_xyt(par[3 * i], par[3 * i + 1], par[3 * i + 2]);
      T = aff_mul(aff_inv(Ttar), Tsrc);
    }
    const cfear_cell* cs = &src->cells[j];
    const cfear_cell* ct = &tar->cells[ti];
    const double nx = T.l0 * cs->normal[0] + T.l1 * cs->normal[1];
    const double ny = T.l2 * cs->normal[0] + T.l3 * cs->normal[1];
    const double sim = fmax(nx * ct->normal[0] + ny * ct->normal[1], 0.0);
    W.w[o] = get_weight(P.weight_opt, (double)cs->nsamples, (double)ct->nsamples, sim, cs->scale, ct->scale);
    W.tmx[o] = (Ttar.l0 * ct->mean[0] + Ttar.l1 * ct->mean[1]) + Ttar.t0;
    W.tmy[o] = (Ttar.l2 * ct->mean[0] + Ttar.l3 * ct->mean[1]) + Ttar.t1;
    W.sx[o] = cs->mean[0]; W.sy[o] = cs->mean[1];
    if (P.cost == CFEAR_COST_P2D) {  // :290-299
      const double a = ct->cov[0], b = ct->cov[1], c = ct->cov[2];
      const double r00 = Ttar.l0, r01 = Ttar.l1, r10 = Ttar.l2, r11 = Ttar.l3;
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
      W.a0[o] = l00; W.a1[o] = l10; W.a2[o] = l11;
    } else {
      W.a0[o] = Ttar.l0 * ct->normal[0] + Ttar.l1 * ct->normal[1];
      W.a1[o] = Ttar.l2 * ct->normal[0] + Ttar.l3 * ct->normal[1];
      W.a2[o] = 0;
    }
    o++;
  }
  __syncthreads();
  return M;
}

// n_scan_normal_reg::Register. poses: n x 3 in global memory (in/out); cov6: 36 doubles or null;
// out: summary in global memory. par_lds: >= 3*n doubles of LDS.
__device__ inline int register_block(ScanDev* const* scans, int n, double* poses, double* cov6, const RegParams& P,
                                     const RegScratch& W, double* par_lds, cfear_reg_summary* out) {
  const int tid = threadIdx.x;
  // Affine3dToVectorXYeZ(Tsrc[i]) (:88-92): theta -> atan2(sin, cos)
  for (int i = tid; i < n; i += blockDim.x) {
    const Aff2 T = aff_from_xyt(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]);
    double v[3]; aff_to_xyt(T, v);
    par_lds[3 * i] = v[0]; par_lds[3 * i + 1] = v[1]; par_lds[3 * i + 2] = v[2];
  }
  if (tid == 0 && out) {
    out->success = 0; out->usable = 0; out->outer_iterations = 0; out->num_residuals = 0; out->num_residual_blocks = 0;
    out->reserved = 0; out->final_cost = 0; out->score = 0;
    for (int i = 0; i < CFEAR_MAX_OUTER; i++) { out->inner_iterations[i] = 0; out->termination[i] = 0; out->outer_cost[i] = 0; out->outer_pose[i][0] = out->outer_pose[i][1] = out->outer_pose[i][2] = 0; }
  }
  __syncthreads();
  const int L = 3 * (n - 1);
  double x[3] = {par_lds[L], par_lds[L + 1], par_lds[L + 2]};
  double tsrc_last[3] = {poses[L], poses[L + 1], poses[L + 2]};
  bool success = true;
  double prev_par[3] = {x[0], x[1], x[2]};
  double prev_score = 1.7976931348623157e308;
  const int rpb = (P.cost == CFEAR_COST_P2L) ? 1 : 2;
  int M = 0, nres = 0;
  SolveSummary ss; ss.num_iterations = 0; ss.termination = 0; ss.final_cost = 0; ss.last_relative_decrease = 0;
  int itr;
  for (itr = 1; itr <= P.max_outer && success; itr++) {  // :102
    __syncthreads();
    if (tid == 0) { par_lds[L] = x[0]; par_lds[L + 1] = x[1]; par_lds[L + 2] = x[2]; }
    __syncthreads();
    M = build_problem_block(scans, n, par_lds, P, itr, W);
    nres = M * rpb;
    if (nres <= 1) { success = false; break; }  // :370-371
    ss = lm_solve_block(W, M, P, x);
    success = (ss.termination != 2);
    if (success) { tsrc_last[0] = x[0]; tsrc_last[1] = x[1]; tsrc_last[2] = x[2]; }
    if (tid == 0 && out && itr - 1 < CFEAR_MAX_OUTER) {
      out->inner_iterations[itr - 1] = ss.num_iterations; out->termination[itr - 1] = ss.termination;
      out->outer_cost[itr - 1] = ss.final_cost;
      out->outer_pose[itr - 1][0] = x[0]; out->outer_pose[itr - 1][1] = x[1]; out->outer_pose[itr - 1][2] = x[2];
    }
    const double current_score = ss.final_cost;
    const double rel_improvement = (prev_score - current_score) / prev_score;
    if (itr > P.min_itr) {  // :134-149
      if (prev_score < current_score) { x[0] = prev_par[0]; x[1] = prev_par[1]; x[2] = prev_par[2]; break; }
      else if (rel_improvement < 0.00001) break;
      else if (ss.last_relative_decrease < 0.00001 || ss.num_iterations == 1) break;
    }
    prev_score = current_score;
    prev_par[0] = x[0]; prev_par[1] = x[1]; prev_par[2] = x[2];
  }
  int ret = 0;
  if (success) {
    // GetCovariance (:392-433): (J~^T J~)^-1 of the last built problem at the final parameters
    const NormalEq E = evaluate_block(W, M, P, x[0], x[1], x[2]);
    const double a = E.h00, b = E.h01, c = E.h02, d = E.h11, e = E.h12, f = E.h22;
    const double C00 = d * f - e * e, C01 = c * e - b * f, C02 = b * e - c * d;
    const double det = a * C00 + b * C01 + c * C02;
    const int dof = nres - 3;
    const bool ok = det > 0 && isfinite(det) && dof != 0;
    ret = ok ? 1 : 0;
    __syncthreads();
    if (tid == 0) {
      for (int i = 0; i < n; i++) { poses[3 * i] = par_lds[3 * i]; poses[3 * i + 1] = par_lds[3 * i + 1]; poses[3 * i + 2] = par_lds[3 * i + 2]; }
      poses[L] = x[0]; poses[L + 1] = x[1]; poses[L + 2] = x[2];
      if (cov6) {
        for (int i = 0; i < 36; i++) cov6[i] = 0;
        cov6[0] = 0.1 * 0.1; cov6[7] = 0.1 * 0.1; cov6[35] = 0.01 * 0.01;  // :173
        if (ok) {
          const double sc = 30 * (ss.final_cost / dof) / det;
          for (int i = 0; i < 36; i++) cov6[i] = (i % 7 == 0) ? 1.0 : 0.0;
          cov6[0] = sc * C00; cov6[1] = sc * C01; cov6[6] = sc * C01; cov6[7] = sc * (a * f - c * c);
          cov6[35] = sc * (a * d - b * b); cov6[5] = sc * C02; cov6[30] = sc * C02;  // (1,5)/(5,1) stay 0 (:426-430)
        }
      }
    }
  } else if (tid == 0) {
    poses[L] = tsrc_last[0]; poses[L + 1] = tsrc_last[1]; poses[L + 2] = tsrc_last[2];
  }
  if (tid == 0 && out) {
    out->success = ret; out->usable = success ? 1 : 0; out->outer_iterations = itr;
    out->num_residuals = nres; out->num_residual_blocks = M; out->final_cost = ss.final_cost;
    out->score = success ? ss.final_cost / nres : 0.0;
  }
  __syncthreads();
  return ret;
}

}  // namespace cfear_dev
