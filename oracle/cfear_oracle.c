/*
 * cfear_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT).
 * See cfear_oracle.h for the scope statement.  PARITY UNPINNED (no reference
 * golden vectors exist; reference unbuildable here: ROS/PCL/Ceres/Eigen/OpenCV absent).
 *
 * Compile with -ffp-contract=off: the reference is built with plain -O3 on x86-64
 * (CMakeLists.txt:4-5,32-33), i.e. without FMA contraction, and several decisions
 * (voxel index, float d^2 < r^2) are rounding sensitive.
 *
 * [3P] marks restatements of third-party behaviour (PCL VoxelGrid, FLANN, Eigen,
 * Ceres) that is not vendored in /root/reference; see SURVEY.md section 9.
 */
#include "cfear_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ---- [3P] sensitivity modes (cfear_oracle.h): deliberate variations of third-party behaviour that the reference's sources do not
 * pin, switched on by tests/run_3p_sensitivity.py only to BOUND their effect on poses and iteration counts. 0 = the oracle. ---- */
static unsigned g_pert = 0;
static uint64_t g_pert_seed = 1;
static cfo_voxel_sorter g_voxel_sorter = NULL;
void cfo_set_perturbation(unsigned mask, uint64_t seed) { g_pert = mask; g_pert_seed = seed ? seed : 1; }
void cfo_set_voxel_sorter(cfo_voxel_sorter fn) { g_voxel_sorter = fn; }
unsigned cfo_get_perturbation(void) { return g_pert; }
static uint64_t pert_rand(uint64_t* st) { /* xorshift64* */
  uint64_t x = *st; x ^= x >> 12; x ^= x << 25; x ^= x >> 27; *st = x;
  return x * 0x2545F4914F6CDD1DULL;
}
/* sum of v[0..n) in the order of the active mode: as written (sequential), reversed, pairwise (a balanced tree), or Eigen 3.3's
 * vectorised redux for an aligned VectorXd with 2-double packets (what `w.sum()` at pointnormal.cpp:19 compiles to on x86-64 /
 * SSE2: two packet accumulators over quads, then the odd packet, the horizontal add, the scalar tail) */
static double pert_sum_tree(const double* v, int n) {
  if (n <= 0) return 0.0;
  if (n == 1) return v[0];
  if (n == 2) return v[0] + v[1];
  const int h = n / 2;
  return pert_sum_tree(v, h) + pert_sum_tree(v + h, n - h);
}
static double pert_sum(const double* v, int n, int is_wsum) {
  if (g_pert & CFO_PERT_SUM_REVERSE) { double s = 0; for (int i = n - 1; i >= 0; i--) s += v[i]; return s; }
  if (g_pert & CFO_PERT_SUM_PAIRWISE) return pert_sum_tree(v, n);
  if ((g_pert & CFO_PERT_WSUM_EIGEN_REDUX) && is_wsum && n >= 2) {
    const int a2 = (n / 2) * 2, a4 = (n / 4) * 4;
    double p0[2] = {v[0], v[1]};
    if (a2 > 2) {
      double p1[2] = {v[2], v[3]};
      for (int i = 4; i < a4; i += 4) { p0[0] += v[i]; p0[1] += v[i + 1]; p1[0] += v[i + 2]; p1[1] += v[i + 3]; }
      p0[0] += p1[0]; p0[1] += p1[1];
      if (a2 > a4) { p0[0] += v[a4]; p0[1] += v[a4 + 1]; }
    }
    double s = p0[0] + p0[1];
    for (int i = a2; i < n; i++) s += v[i];
    return s;
  }
  double s = 0; for (int i = 0; i < n; i++) s += v[i]; return s;
}

static double now_s(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void cfo_default_params(cfo_params* p) {
  memset(p, 0, sizeof(*p));
  p->z_min = 60.f;            /* radar_driver.h:40 */
  p->range_res = 0.0438f;     /* radar_driver.h:41 */
  p->min_distance = 2.5f;     /* radar_driver.h:45 */
  p->k_strongest = 12;        /* radar_driver.h:42 */
  p->res = 3.0;               /* odometrykeyframefuser.h:132 */
  p->downsample_factor = 1.0; /* pointnormal.cpp:5 */
  p->weight_intensity = 1;    /* offline_odometry.cpp:160 */
  p->cost = CFO_COST_P2L;     /* odometrykeyframefuser.h:86 */
  p->loss = CFO_LOSS_HUBER;   /* odometrykeyframefuser.h:99 */
  p->weight_opt = 4;          /* launch/oxford/eval/params/baseline */
  p->loss_limit = 0.1;
  p->covar_scale = 1.0;
  p->regularization = 0.1;
  p->submap_scan_size = 4;
  p->compensate = 1;
  p->radar_ccw = 0;
  p->use_keyframe = 1;
  p->min_keyframe_dist = 1.5;
  p->min_keyframe_rot_deg = 5.0;
  p->max_itr_association = 8; /* n_scan_normal.h:75 */
  p->min_itr = 3;
  p->max_solver_iterations = 20; /* n_scan_normal.cpp:9 */
  p->assoc_radius = 2.0;         /* registration.h:122 */
}

/* ------------------------------------------------------------------------------------------
 * Stage 1: k-strongest (radar_filters.cpp:209-237) and peaks (radar_filters.cpp:238-298)
 * ------------------------------------------------------------------------------------------ */

#define SLOT(range, inten) ((uint32_t)(range) | ((uint32_t)(inten) << 16) | (1u << 24))

/* lexicographic std::pair<uchar,int> operator< */
static int pair_less(int i1, int r1, int i2, int r2) { return (i1 < i2) || (i1 == i2 && r1 < r2); }

int cfo_filter(const uint8_t* img, int A, int R, int z_min, int k, uint32_t* out) {
  if (!img || !out || A <= 0 || R <= 0 || k <= 0 || R > 65536) return -1;
  const uint8_t u_zmin = (uint8_t)z_min; /* radar_filters.cpp:212 */
  int* vi = (int*)malloc(sizeof(int) * (size_t)(k + 2));
  int* vr = (int*)malloc(sizeof(int) * (size_t)(k + 2));
  uint16_t* score = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)(R + 16));
  uint8_t* has = (uint8_t*)malloc((size_t)(R + 16));
  const long total = (long)A * (long)R;
  memset(out, 0, sizeof(uint32_t) * (size_t)A * (size_t)k);
  for (int b = 0; b < A; b++) {
    const uint8_t* row = img + (size_t)b * (size_t)R;
    int n = 0;
    /* radar_filters.cpp:215-229: bounded ascending vector, lower_bound insert, erase begin */
    for (int range = 0; range < R; range++) {
      const int inten = row[range];
      if (inten < u_zmin) continue;
      if (n == 0) {
        vi[0] = inten; vr[0] = range; n = 1;
      } else {
        int lo = 0, hi = n; /* std::lower_bound */
        while (lo < hi) {
          int mid = (lo + hi) / 2;
          if (pair_less(vi[mid], vr[mid], inten, range)) lo = mid + 1; else hi = mid;
        }
        for (int j = n; j > lo; j--) { vi[j] = vi[j - 1]; vr[j] = vr[j - 1]; }
        vi[lo] = inten; vr[lo] = range; n++;
        if (n > k) { /* erase(begin) */
          for (int j = 0; j + 1 < n; j++) { vi[j] = vi[j + 1]; vr[j] = vr[j + 1]; }
          n--;
        }
      }
    }
    uint32_t* o = out + (size_t)b * (size_t)k;
    for (int j = 0; j < n; j++) o[j] = SLOT(vr[j], vi[j]);

    /* AxialNonMaxSupress: std::unordered_map<int,uint16_t> score restated as has[]/score[],
     * index shifted by +8 so that keys -3..R+2 are representable (operator[] default = 0). */
    const int ws = 3;
    memset(has, 0, (size_t)(R + 16));
    memset(score, 0, sizeof(uint16_t) * (size_t)(R + 16));
    for (int j = 0; j < n; j++) {
      const int m = vr[j];
      if (m < ws || m >= R - ws) continue; /* radar_filters.cpp:251 */
      for (int rn = m - ws; rn <= m + ws; rn++) {
        if (!has[rn + 8]) {
          uint16_t s = 0;
          for (int rnn = rn - ws; rnn <= rn + ws; rnn++) {
            /* cv::Mat::at<uchar>(bearing, r_nn) unchecked: address = data + bearing*step + r_nn
             * (radar_filters.cpp:260). Inside the image buffer this reads the neighbouring row;
             * outside it is UB in the reference -- defined here as 0. */
            const long off = (long)b * (long)R + (long)rnn;
            const uint8_t v = (off >= 0 && off < total) ? img[off] : 0;
            s = (uint16_t)(s + (uint16_t)v);
          }
          score[rn + 8] = s; has[rn + 8] = 1;
        }
      }
    }
    for (int j = 0; j < n; j++) {
      const int m = vr[j];
      int largest = 1;
      const uint16_t pthis = score[m + 8];
      for (int i = 1; i <= ws; i++) {
        const uint16_t pnext = (m + i + 8 < R + 16) ? score[m + i + 8] : 0;
        const uint16_t pprev = (m - i + 8 >= 0) ? score[m - i + 8] : 0;
        if (pprev > pthis || pthis < pnext) { largest = 0; break; } /* radar_filters.cpp:282 */
      }
      if (largest) o[j] |= (1u << 25);
    }
  }
  free(vi); free(vr); free(score); free(has);
  return 0;
}

typedef struct { int inten; int range; } ir_t;
static int ir_cmp_desc(const void* a, const void* b) {
  const ir_t* x = (const ir_t*)a; const ir_t* y = (const ir_t*)b;
  if (x->inten != y->inten) return y->inten - x->inten;
  return y->range - x->range;
}
/* Independent statement of the selection rule: the k largest by key (I, range),
 * emitted ascending. No peak flag. */
int cfo_filter_bruteforce(const uint8_t* img, int A, int R, int z_min, int k, uint32_t* out) {
  if (!img || !out || A <= 0 || R <= 0 || k <= 0) return -1;
  const uint8_t u_zmin = (uint8_t)z_min;
  ir_t* v = (ir_t*)malloc(sizeof(ir_t) * (size_t)R);
  memset(out, 0, sizeof(uint32_t) * (size_t)A * (size_t)k);
  for (int b = 0; b < A; b++) {
    int n = 0;
    for (int r = 0; r < R; r++) {
      int I = img[(size_t)b * R + r];
      if (I >= u_zmin) { v[n].inten = I; v[n].range = r; n++; }
    }
    qsort(v, (size_t)n, sizeof(ir_t), ir_cmp_desc);
    int m = n < k ? n : k;
    for (int j = 0; j < m; j++) out[(size_t)b * k + j] = SLOT(v[m - 1 - j].range, v[m - 1 - j].inten);
  }
  free(v);
  return 0;
}

/* getPeaksFilteredPointCloud (radar_filters.cpp:309-337) */
int cfo_cloud(const uint32_t* slots, int A, int k, float range_res_f, float min_distance_f, int peaks,
              float* xyi) {
  const double range_res = (double)range_res_f;       /* float -> double at radar_driver.cpp:58 */
  const double min_distance = (double)min_distance_f;
  const int min_range_bin = (int)ceil(min_distance / range_res); /* :315 */
  int n = 0;
  for (int b = 0; b < A; b++) {
    const double theta = ((double)(b + 1) / A) * 2. * M_PI; /* :317 */
    const double cos_t = cos(theta), sin_t = sin(theta);
    const double range_res_half = range_res / 2.0;
    for (int j = 0; j < k; j++) {
      const uint32_t s = slots[(size_t)b * k + j];
      if (!CFO_SLOT_VALID(s)) continue;
      if (peaks && !CFO_SLOT_PEAK(s)) continue;
      const int range = CFO_SLOT_RANGE(s);
      if (range > min_range_bin) { /* :327 */
        xyi[3 * n + 0] = (float)((range_res_half + range_res * range) * cos_t);
        xyi[3 * n + 1] = (float)((range_res_half + range_res * range) * sin_t);
        xyi[3 * n + 2] = (float)CFO_SLOT_INTENSITY(s);
        n++;
      }
    }
  }
  return n;
}

/* ------------------------------------------------------------------------------------------
 * Stage 1, alternative filter: azimuth CA-CFAR (cfar.cpp:12-16, 27-87; selected by
 * filter_type "CA-CFAR", radar_driver.cpp:52-56 with max_distance = 400.0)
 * ------------------------------------------------------------------------------------------ */
/* CFARFilter::getCAScalingFactor (cfar.cpp:12-16) for the two windows together (cfar.cpp:32) */
double cfo_cfar_scaling(int window_size, double false_alarm_rate) {
  const double N = (double)(window_size * 2);
  return N * (pow(false_alarm_rate, -1. / N) - 1.);
}

/* AzimuthCACFAR::getMean (cfar.cpp:76-86): mean of the squared intensities of [start, end). The reference runs a
 * size_t index up to an int bound; a negative bound (range_bin < nb_guard_cells, only reachable when min_distance
 * is below the guard distance) is undefined behaviour there and an empty window here. Empty window: 0/0 = NaN. */
static double cfar_mean(const uint8_t* row, int start, int end) {
  double sum = 0., N = 0.;
  for (int i = start; i < end; i++) { sum += pow((double)row[i], 2.); N += 1.; }
  return sum / N;
}

/* AzimuthCACFAR::getFilteredPointCloud (cfar.cpp:35-74). Returns the number of detections; the first `cap` are
 * written to xyi (x, y, intensity), row-major over (azimuth, range bin). */
int cfo_cfar(const uint8_t* img, int A, int R, float range_res_f, float static_threshold_f, float min_distance_f,
             double max_distance, int window_size, int nb_guard_cells, float false_alarm_rate_f, float* xyi, int cap) {
  /* float members of radarDriver::Parameters bound to const double& (radar_driver.cpp:54) */
  const double range_resolution = (double)range_res_f, static_threshold = (double)static_threshold_f;
  const double min_distance = (double)min_distance_f;
  const double scaling_factor = cfo_cfar_scaling(window_size, (double)false_alarm_rate_f);
  int n = 0;
  for (int az = 0; az < A; az++) {
    const uint8_t* row = img + (size_t)az * R;
    const double theta = ((double)(az + 1) / A) * 2. * M_PI; /* :40 */
    for (int bin = 0; bin < R; bin++) {
      const double range = range_resolution * (double)bin;
      const double intensity = (double)row[bin];
      if (range > min_distance && range < max_distance && intensity > static_threshold) { /* :45 */
        const int t0 = bin - nb_guard_cells - window_size > 0 ? bin - nb_guard_cells - window_size : 0; /* :48 */
        const int t1 = bin - nb_guard_cells;
        const double trailing_mean = cfar_mean(row, t0, t1);
        const int f0 = bin + nb_guard_cells; /* :52 */
        const int f1 = R < bin + nb_guard_cells + window_size ? R : bin + nb_guard_cells + window_size;
        const double forwarding_mean = cfar_mean(row, f0, f1);
        const double mean = (trailing_mean + forwarding_mean) / 2.0; /* :56 */
        const double threshold = scaling_factor * mean;
        const double squared_intensity = pow(intensity, 2.);
        if (squared_intensity > threshold) { /* :60 (false for NaN) */
          if (n < cap) {
            xyi[3 * n + 0] = (float)(range * cos(theta));
            xyi[3 * n + 1] = (float)(range * sin(theta));
            xyi[3 * n + 2] = (float)intensity;
          }
          n++;
        }
      }
    }
  }
  return n;
}

/* The same detector with the window sums taken off a prefix sum of squares instead of being re-added per bin. Not a different arithmetic: every
 * partial sum the reference's getMean forms is an integer below 2^53, so its double additions are exact and `sum` there equals the integer window sum
 * here; N, the division, the mean of the two means, the scaling and the comparison are the statements above. (For the long windows of the reference's
 * sweep - launch/oxford/eval/params/kstrong_vs_cfar/oxford-cfear-3-ca-cfar:31: up to 500 bins - the literal version spends a second per sweep;
 * tests/test_oracle_cpu.py checks the two against each other bin for bin.) */
int cfo_cfar_prefix(const uint8_t* img, int A, int R, float range_res_f, float static_threshold_f, float min_distance_f,
                    double max_distance, int window_size, int nb_guard_cells, float false_alarm_rate_f, float* xyi, int cap) {
  const double range_resolution = (double)range_res_f, static_threshold = (double)static_threshold_f;
  const double min_distance = (double)min_distance_f;
  const double scaling_factor = cfo_cfar_scaling(window_size, (double)false_alarm_rate_f);
  unsigned long long* pre = (unsigned long long*)malloc(sizeof(unsigned long long) * ((size_t)R + 1));
  int n = 0;
  for (int az = 0; az < A; az++) {
    const uint8_t* row = img + (size_t)az * R;
    const double theta = ((double)(az + 1) / A) * 2. * M_PI;
    pre[0] = 0;
    for (int i = 0; i < R; i++) pre[i + 1] = pre[i] + (unsigned long long)row[i] * row[i];
    for (int bin = 0; bin < R; bin++) {
      const double range = range_resolution * (double)bin;
      const double intensity = (double)row[bin];
      if (range > min_distance && range < max_distance && intensity > static_threshold) {
        const int t0 = bin - nb_guard_cells - window_size > 0 ? bin - nb_guard_cells - window_size : 0;
        const int t1 = bin - nb_guard_cells;
        const int f0 = bin + nb_guard_cells;
        const int f1 = R < bin + nb_guard_cells + window_size ? R : bin + nb_guard_cells + window_size;
        const double tsum = t1 > t0 ? (double)(pre[t1] - pre[t0]) : 0., tN = t1 > t0 ? (double)(t1 - t0) : 0.;
        const double fsum = f1 > f0 ? (double)(pre[f1] - pre[f0]) : 0., fN = f1 > f0 ? (double)(f1 - f0) : 0.;
        const double mean = (tsum / tN + fsum / fN) / 2.0;
        const double threshold = scaling_factor * mean;
        const double squared_intensity = pow(intensity, 2.);
        if (squared_intensity > threshold) {
          if (n < cap) {
            xyi[3 * n + 0] = (float)(range * cos(theta));
            xyi[3 * n + 1] = (float)(range * sin(theta));
            xyi[3 * n + 2] = (float)intensity;
          }
          n++;
        }
      }
    }
  }
  free(pre);
  return n;
}

/* utils.h:28-32 */
static double rel_time_stamp(double x, double y, int ccw) {
  double a = atan2(y, x);
  double d = ((a > 0.00001 ? a : (2 * M_PI + a)) / (2 * M_PI));
  return ccw ? -(d - 0.5) : (d - 0.5);
}

/* utils.cpp:96-107 (+ getScaledRotationMatrix/TranslationVector :130-146) */
void cfo_compensate(float* xyi, int n, const double mot[3], int ccw) {
  for (int i = 0; i < n; i++) {
    const double px = xyi[3 * i], py = xyi[3 * i + 1];
    const double d = rel_time_stamp(px, py, ccw);
    const double s1 = sin(d * mot[2]), c1 = cos(d * mot[2]);
    const double tx = d * mot[0], ty = d * mot[1];
    xyi[3 * i + 0] = (float)((c1 * px + (-s1) * py) + tx);
    xyi[3 * i + 1] = (float)((s1 * px + c1 * py) + ty);
  }
}

/* ------------------------------------------------------------------------------------------
 * Stage 2: oriented surface points (pointnormal.cpp:7-63, 65-90, 151-162, 238-254, 265-297)
 * ------------------------------------------------------------------------------------------ */

struct cfo_scan {
  int n;          /* input_ size */
  float* pts;     /* x,y,intensity */
  int nsamp;
  float* samples; /* voxel centroids */
  int ncells;
  cfo_cell* cells;
  float* mean_f;  /* downsampled_: float(u_) (pointnormal.cpp:151-158) */
  /* uniform grid over mean_f (acceleration only; result == brute force) */
  double gminx, gminy, gcell;
  int gw, gh;
  int* gstart;
  int* gorder;
  struct cfo_kdtree* kd; /* CFO_PERT_NN_TIE_FLANN: the kd-tree over mean_f, built on first use */
};

typedef struct { uint64_t key; } vkey_t;
static int u64_cmp(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return (x > y) - (x < y);
}
typedef struct { float d2; int idx; } nb_t;
static int nb_cmp(const void* a, const void* b) {
  const nb_t* x = (const nb_t*)a; const nb_t* y = (const nb_t*)b;
  if (x->d2 < y->d2) return -1;
  if (x->d2 > y->d2) return 1;
  return (x->idx > y->idx) - (x->idx < y->idx); /* [3P] flann DistanceIndex operator< */
}

/* Symmetric 2x2 eigen-decomposition, closed form. [3P] Eigen::SelfAdjointEigenSolver<Matrix2d>
 * (pointnormal.cpp:39-45) is iterative; equal up to rounding, eigenvalues ascending, unit vectors,
 * identity eigenvectors for an isotropic matrix. Reads the lower triangle like Eigen. */
/* the same decomposition by one Jacobi rotation (Rutishauser's stable tangent): a different sequence of roundings for the same
 * mathematical result - stands in for Eigen's iterative tridiagonal QR, whose last ulps differ from the closed form too */
static void eig2_jacobi(double a, double b, double c, double* lmin, double* lmax, double vmin[2], double vmax[2]) {
  double l0 = a, l1 = c, cs = 1.0, sn = 0.0;
  if (b != 0.0) {
    const double theta = (c - a) / (2.0 * b);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    cs = 1.0 / sqrt(t * t + 1.0); sn = t * cs;
    l0 = a - t * b; l1 = c + t * b;
  }
  /* eigenvectors: (cs, -sn) for l0, (sn, cs) for l1 */
  if (l0 <= l1) { *lmin = l0; *lmax = l1; vmin[0] = cs; vmin[1] = -sn; vmax[0] = sn; vmax[1] = cs; }
  else { *lmin = l1; *lmax = l0; vmin[0] = sn; vmin[1] = cs; vmax[0] = cs; vmax[1] = -sn; }
}
static void eig2(double a, double b, double c, double* lmin, double* lmax, double vmin[2], double vmax[2]) {
  if (g_pert & CFO_PERT_EIG_JACOBI) { eig2_jacobi(a, b, c, lmin, lmax, vmin, vmax); return; }
  const double t1 = 0.5 * (a + c);
  const double d = 0.5 * (a - c);
  const double t0 = sqrt(d * d + b * b);
  *lmin = t1 - t0;
  *lmax = t1 + t0;
  double v0x = *lmax - c, v0y = b;
  double v1x = b, v1y = *lmax - a;
  const double n0 = v0x * v0x + v0y * v0y, n1 = v1x * v1x + v1y * v1y;
  double vx, vy, nn;
  if (n0 >= n1) { vx = v0x; vy = v0y; nn = n0; } else { vx = v1x; vy = v1y; nn = n1; }
  if (!(nn > 0.0)) { vmax[0] = 0; vmax[1] = 1; vmin[0] = 1; vmin[1] = 0; return; }
  const double inv = 1.0 / sqrt(nn);
  vmax[0] = vx * inv; vmax[1] = vy * inv;
  vmin[0] = -vmax[1]; vmin[1] = vmax[0];
}

/* cell::cell + cell::ComputeNormal (pointnormal.cpp:7-63); idx = neighbour indices in search order */
static void make_cell(const float* pts, const int* idx, int N, int weight_intensity, cfo_cell* c) {
  memset(c, 0, sizeof(*c));
  c->nsamples = N;
  double sum = 0, ux = 0, uy = 0, cxx = 0, cyx = 0, cyy = 0;
  if (g_pert & (CFO_PERT_SUM_REVERSE | CFO_PERT_SUM_PAIRWISE | CFO_PERT_WSUM_EIGEN_REDUX)) {
    /* [3P] sensitivity modes: the same terms, added up in another order (pert_sum) */
    double* t = (double*)malloc(sizeof(double) * 3 * (size_t)N);
    for (int i = 0; i < N; i++) t[i] = weight_intensity ? fmax((double)pts[3 * idx[i] + 2] - 60.0, 0.0) : 1.0;
    sum = pert_sum(t, N, 1);
    for (int i = 0; i < N; i++) { const double w = (weight_intensity ? fmax((double)pts[3 * idx[i] + 2] - 60.0, 0.0) : 1.0) / sum; t[i] = w * (double)pts[3 * idx[i]]; t[N + i] = w * (double)pts[3 * idx[i] + 1]; }
    ux = pert_sum(t, N, 0); uy = pert_sum(t + N, N, 0);
    for (int i = 0; i < N; i++) {
      const double w = (weight_intensity ? fmax((double)pts[3 * idx[i] + 2] - 60.0, 0.0) : 1.0) / sum;
      const double dx = (double)pts[3 * idx[i]] - ux, dy = (double)pts[3 * idx[i] + 1] - uy;
      t[i] = dx * (w * dx); t[N + i] = dy * (w * dx); t[2 * N + i] = dy * (w * dy);
    }
    cxx = pert_sum(t, N, 0); cyx = pert_sum(t + N, N, 0); cyy = pert_sum(t + 2 * N, N, 0);
    free(t);
  } else {
  for (int i = 0; i < N; i++) { /* :13-18 */
    const double w = weight_intensity ? fmax((double)pts[3 * idx[i] + 2] - 60.0, 0.0) : 1.0;
    sum += w;
  }
  for (int i = 0; i < N; i++) { /* :21-24 */
    const double w = (weight_intensity ? fmax((double)pts[3 * idx[i] + 2] - 60.0, 0.0) : 1.0) / sum;
    ux += w * (double)pts[3 * idx[i]];
    uy += w * (double)pts[3 * idx[i] + 1];
  }
  for (int i = 0; i < N; i++) { /* :26-33: cov = x^T * (w .* x) */
    const double w = (weight_intensity ? fmax((double)pts[3 * idx[i] + 2] - 60.0, 0.0) : 1.0) / sum;
    const double dx = (double)pts[3 * idx[i]] - ux, dy = (double)pts[3 * idx[i] + 1] - uy;
    cxx += dx * (w * dx);
    cyx += dy * (w * dx); /* lower triangle entry (1,0), the one the eigensolver reads */
    cyy += dy * (w * dy);
  }
  }
  c->sum_intensity = sum;
  c->avg_intensity = sum / N;
  c->mean[0] = ux; c->mean[1] = uy;
  c->cov[0] = cxx; c->cov[1] = cyx; c->cov[2] = cyy;
  double lmin, lmax, vmin[2], vmax[2];
  eig2(cxx, cyx, cyy, &lmin, &lmax, vmin, vmax);
  c->lambda_min = lmin; c->lambda_max = lmax;
  const double cond = fabs(lmax / lmin);      /* :53 */
  const double det = lmax * lmin;             /* :54 */
  c->valid = (cond <= 10000) && (det > 0.00001) && lmin > 0 && lmax > 0; /* :56 */
  c->scale = log(1.0 + cond / 2);             /* :57 */
  /* origin = (0,0) (odometrykeyframefuser.cpp:161): Po_u = origin - u */
  if (vmin[0] * (0.0 - ux) + vmin[1] * (0.0 - uy) < 0) { vmin[0] = -vmin[0]; vmin[1] = -vmin[1]; } /* :59-61 */
  c->normal[0] = vmin[0]; c->normal[1] = vmin[1];
  c->orth[0] = vmax[0]; c->orth[1] = vmax[1];
}

static int lower_bound_u64(const uint64_t* a, int n, uint64_t v) {
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) / 2; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}

static void build_cell_grid(cfo_scan* s, double cell) {
  s->gcell = cell; s->gw = s->gh = 0; s->gstart = NULL; s->gorder = NULL;
  const int n = s->ncells;
  if (n == 0) return;
  double minx = s->mean_f[0], maxx = minx, miny = s->mean_f[1], maxy = miny;
  for (int i = 1; i < n; i++) {
    double x = s->mean_f[2 * i], y = s->mean_f[2 * i + 1];
    if (x < minx) minx = x; if (x > maxx) maxx = x;
    if (y < miny) miny = y; if (y > maxy) maxy = y;
  }
  s->gminx = minx; s->gminy = miny;
  s->gw = (int)floor((maxx - minx) / cell) + 1;
  s->gh = (int)floor((maxy - miny) / cell) + 1;
  const int G = s->gw * s->gh;
  s->gstart = (int*)calloc((size_t)G + 1, sizeof(int));
  s->gorder = (int*)malloc(sizeof(int) * (size_t)n);
  int* gi = (int*)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; i++) {
    int gx = (int)floor((s->mean_f[2 * i] - minx) / cell), gy = (int)floor((s->mean_f[2 * i + 1] - miny) / cell);
    gi[i] = gy * s->gw + gx;
    s->gstart[gi[i] + 1]++;
  }
  for (int g = 0; g < G; g++) s->gstart[g + 1] += s->gstart[g];
  int* fill = (int*)malloc(sizeof(int) * (size_t)G);
  memcpy(fill, s->gstart, sizeof(int) * (size_t)G);
  for (int i = 0; i < n; i++) s->gorder[fill[gi[i]]++] = i; /* ascending cell index inside a bucket */
  free(fill); free(gi);
}

cfo_scan* cfo_scan_create(const float* xyi, int n, const cfo_params* p, int brute) {
  if (n <= 0 || !xyi) return NULL; /* reference: exit(0) on an empty cloud (pointnormal.cpp:72-75) */
  cfo_scan* s = (cfo_scan*)calloc(1, sizeof(cfo_scan));
  s->n = n;
  s->pts = (float*)malloc(sizeof(float) * 3 * (size_t)n);
  memcpy(s->pts, xyi, sizeof(float) * 3 * (size_t)n);
  const float radius = (float)p->res; /* pointnormal.h:118: float radius */
  /* [3P] pcl::VoxelGrid<PointXYZI>::applyFilter, leaf = radius_/downsample_factor (pointnormal.cpp:279) */
  const float leaf = (float)((double)radius / p->downsample_factor);
  const float inv = 1.0f / leaf;
  float minx = xyi[0], maxx = xyi[0], miny = xyi[1], maxy = xyi[1];
  for (int i = 1; i < n; i++) {
    const float x = xyi[3 * i], y = xyi[3 * i + 1];
    if (x < minx) minx = x; if (x > maxx) maxx = x;
    if (y < miny) miny = y; if (y > maxy) maxy = y;
  }
  const int min_b0 = (int)floorf(minx * inv), max_b0 = (int)floorf(maxx * inv);
  const int min_b1 = (int)floorf(miny * inv), max_b1 = (int)floorf(maxy * inv);
  const int div0 = max_b0 - min_b0 + 1, div1 = max_b1 - min_b1 + 1;
  uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  for (int i = 0; i < n; i++) {
    const int ijk0 = (int)(floorf(xyi[3 * i] * inv) - (float)min_b0);
    const int ijk1 = (int)(floorf(xyi[3 * i + 1] * inv) - (float)min_b1);
    const uint64_t idx = (uint64_t)((int64_t)ijk0 + (int64_t)ijk1 * (int64_t)div0); /* z == 0 -> ijk2 = 0 */
    keys[i] = (idx << 24) | (uint64_t)i; /* [3P] std::sort on idx only is unstable; pinned here as stable */
  }
  qsort(keys, (size_t)n, sizeof(uint64_t), u64_cmp);
  if (g_pert & (CFO_PERT_VOXEL_REVERSE | CFO_PERT_VOXEL_RANDOM | CFO_PERT_VOXEL_STDSORT)) {
    /* [3P] sensitivity modes: another order of the points INSIDE a voxel (PCL sorts on the voxel index only, with an unstable
     * sort: std::sort up to 1.9, boost's integer_sort from 1.10) - the float centroid sums below then round differently */
    if ((g_pert & CFO_PERT_VOXEL_STDSORT) && g_voxel_sorter) {
      /* exactly PCL <= 1.9: index_vector in point order, std::sort with operator< on idx (oracle/stdsort_perm.cpp, libstdc++) */
      uint32_t* vi = (uint32_t*)malloc(sizeof(uint32_t) * 2 * (size_t)n);
      uint32_t* pi = vi + n;
      for (int i = 0; i < n; i++) {
        const int ijk0 = (int)(floorf(xyi[3 * i] * inv) - (float)min_b0);
        const int ijk1 = (int)(floorf(xyi[3 * i + 1] * inv) - (float)min_b1);
        vi[i] = (uint32_t)(ijk0 + ijk1 * div0); pi[i] = (uint32_t)i;
      }
      g_voxel_sorter(vi, pi, n);
      for (int i = 0; i < n; i++) keys[i] = ((uint64_t)vi[i] << 24) | (uint64_t)pi[i];
      free(vi);
    } else {
      uint64_t st = g_pert_seed * 0x9E3779B97F4A7C15ULL + (uint64_t)n;
      for (int i = 0; i < n;) {
        int j = i;
        while (j < n && (keys[j] >> 24) == (keys[i] >> 24)) j++;
        if (g_pert & CFO_PERT_VOXEL_REVERSE) {
          for (int a = i, b = j - 1; a < b; a++, b--) { const uint64_t t = keys[a]; keys[a] = keys[b]; keys[b] = t; }
        } else {
          for (int a = j - 1; a > i; a--) { const int b = i + (int)(pert_rand(&st) % (uint64_t)(a - i + 1)); const uint64_t t = keys[a]; keys[a] = keys[b]; keys[b] = t; }
        }
        i = j;
      }
    }
  }
  /* centroids: [3P] pcl::CentroidPoint<PointXYZI>: float sums divided by float(count), ascending idx */
  s->samples = (float*)malloc(sizeof(float) * 3 * (size_t)n);
  int* vstart = (int*)malloc(sizeof(int) * ((size_t)n + 1));
  uint64_t* vidx = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)n);
  int nv = 0;
  for (int i = 0; i < n;) {
    int j = i;
    float sx = 0, sy = 0, si = 0;
    const uint64_t id = keys[i] >> 24;
    while (j < n && (keys[j] >> 24) == id) {
      const int pi = (int)(keys[j] & 0xFFFFFF);
      sx += xyi[3 * pi]; sy += xyi[3 * pi + 1]; si += xyi[3 * pi + 2];
      j++;
    }
    const float cnt = (float)(j - i);
    s->samples[3 * nv] = sx / cnt; s->samples[3 * nv + 1] = sy / cnt; s->samples[3 * nv + 2] = si / cnt;
    vstart[nv] = i; vidx[nv] = id; nv++;
    i = j;
  }
  vstart[nv] = n;
  s->nsamp = nv;
  /* radius search per sample point (pointnormal.cpp:286-296). [3P] flann L2_Simple<float> over (x,y,z=0),
   * accept d2 < float(radius*radius), results sorted ascending by (d2, index). */
  const float r2 = (float)((double)radius * (double)radius);
  s->cells = (cfo_cell*)malloc(sizeof(cfo_cell) * (size_t)(nv > 0 ? nv : 1));
  nb_t* nb = (nb_t*)malloc(sizeof(nb_t) * (size_t)n);
  int* nidx = (int*)malloc(sizeof(int) * (size_t)n);
  const int reach = (int)ceil((double)radius / (double)leaf) + 1;
  for (int v = 0; v < nv; v++) {
    const float cx = s->samples[3 * v], cy = s->samples[3 * v + 1];
    int m = 0;
    if (brute) {
      for (int i = 0; i < n; i++) {
        const float dx = cx - xyi[3 * i], dy = cy - xyi[3 * i + 1];
        float d2 = dx * dx; d2 += dy * dy; d2 += 0.0f * 0.0f;
        if (d2 < r2) { nb[m].d2 = d2; nb[m].idx = i; m++; }
      }
    } else {
      const int vx = (int)(floorf(cx * inv) - (float)min_b0), vy = (int)(floorf(cy * inv) - (float)min_b1);
      for (int gy = vy - reach; gy <= vy + reach; gy++) {
        if (gy < 0 || gy >= div1) continue;
        int gx0 = vx - reach, gx1 = vx + reach;
        if (gx0 < 0) gx0 = 0; if (gx1 >= div0) gx1 = div0 - 1;
        if (gx0 > gx1) continue;
        const uint64_t k0 = (uint64_t)((int64_t)gx0 + (int64_t)gy * div0), k1 = (uint64_t)((int64_t)gx1 + (int64_t)gy * div0);
        const int a = lower_bound_u64(vidx, nv, k0), b = lower_bound_u64(vidx, nv, k1 + 1);
        for (int q = vstart[a]; q < vstart[b]; q++) {
          const int i = (int)(keys[q] & 0xFFFFFF);
          const float dx = cx - xyi[3 * i], dy = cy - xyi[3 * i + 1];
          float d2 = dx * dx; d2 += dy * dy; d2 += 0.0f * 0.0f;
          if (d2 < r2) { nb[m].d2 = d2; nb[m].idx = i; m++; }
        }
      }
    }
    if (m >= 6) { /* pointnormal.cpp:291 */
      qsort(nb, (size_t)m, sizeof(nb_t), nb_cmp);
      for (int i = 0; i < m; i++) nidx[i] = nb[i].idx;
      cfo_cell c;
      make_cell(xyi, nidx, m, p->weight_intensity, &c);
      if (c.valid) s->cells[s->ncells++] = c; /* :293-294 */
    }
  }
  free(nb); free(nidx); free(keys); free(vstart); free(vidx);
  /* ComputeSearchTreeFromCells (pointnormal.cpp:151-162) */
  s->mean_f = (float*)malloc(sizeof(float) * 2 * (size_t)(s->ncells > 0 ? s->ncells : 1));
  for (int i = 0; i < s->ncells; i++) {
    s->mean_f[2 * i] = (float)s->cells[i].mean[0];
    s->mean_f[2 * i + 1] = (float)s->cells[i].mean[1];
  }
  build_cell_grid(s, 2.0 * p->assoc_radius);
  return s;
}

/* ---- [3P-recall] flann::KDTreeSingleIndex<L2_Simple<float>> over 2-D float points, as pcl::KdTreeFLANN<pcl::PointXY> builds it
 * (KDTreeSingleIndexParams(15): leaf_max_size 15, reorder = true) and searches it (nearestKSearch(k = 1): SearchParams(-1, eps 0),
 * KNNSimpleResultSet). Restated from the published algorithm (flann/algorithms/kdtree_single_index.h, FLANN 1.8 / 1.9: divideTree,
 * middleSplit_, planeSplit, computeInitialDistances, searchLevel; flann/util/result_set.h: KNNSimpleResultSet::addPoint), not from a
 * source under /root/reference - the library is not vendored there. What it is for: exact-distance ties. A point replaces the best
 * so far only if strictly nearer, so of several equidistant points the FIRST ONE VISITED is returned; the visiting order is the
 * best-child-first descent over this tree. ---------------------------------------------------------------------------------------- */
typedef struct { float low, high; } kd_interval;
typedef struct { int left, right, divfeat, child1, child2; float divlow, divhigh; } kd_node; /* child1 < 0: leaf over vind[left, right) */
typedef struct cfo_kdtree {
  int n, nnodes, cap_nodes, root;
  int* vind;      /* the permutation divideTree leaves behind */
  float* data;    /* reordered copy: point vind[i] at row i (reorder_) */
  const float* pts;
  kd_node* nodes;
  kd_interval root_bbox[2];
} cfo_kdtree;
static void kd_free(cfo_kdtree* t) { if (t) { free(t->vind); free(t->data); free(t->nodes); free(t); } }
static void kd_minmax(const cfo_kdtree* t, const int* ind, int count, int dim, float* mn, float* mx) { /* computeMinMax */
  *mn = t->pts[2 * ind[0] + dim]; *mx = *mn;
  for (int i = 1; i < count; i++) {
    const float v = t->pts[2 * ind[i] + dim];
    if (v < *mn) *mn = v;
    if (v > *mx) *mx = v;
  }
}
static void kd_plane_split(const cfo_kdtree* t, int* ind, int count, int cutfeat, float cutval, int* lim1, int* lim2) { /* planeSplit */
  int left = 0, right = count - 1;
  for (;;) {
    while (left <= right && t->pts[2 * ind[left] + cutfeat] < cutval) ++left;
    while (left <= right && t->pts[2 * ind[right] + cutfeat] >= cutval) --right;
    if (left > right) break;
    { const int x = ind[left]; ind[left] = ind[right]; ind[right] = x; } ++left; --right;
  }
  *lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && t->pts[2 * ind[left] + cutfeat] <= cutval) ++left;
    while (left <= right && t->pts[2 * ind[right] + cutfeat] > cutval) --right;
    if (left > right) break;
    { const int x = ind[left]; ind[left] = ind[right]; ind[right] = x; } ++left; --right;
  }
  *lim2 = left;
}
static void kd_middle_split(const cfo_kdtree* t, int* ind, int count, int* index, int* cutfeat, float* cutval, const kd_interval* bbox) { /* middleSplit_ */
  const float EPS = 0.00001f;
  float max_span = bbox[0].high - bbox[0].low;
  for (int i = 1; i < 2; i++) { const float span = bbox[i].high - bbox[i].low; if (span > max_span) max_span = span; }
  float max_spread = -1;
  *cutfeat = 0;
  for (int i = 0; i < 2; i++) {
    const float span = bbox[i].high - bbox[i].low;
    if (span > (float)((1 - EPS) * max_span)) {
      float mn, mx;
      kd_minmax(t, ind, count, i, &mn, &mx);
      const float spread = (float)(mx - mn);
      if (spread > max_spread) { *cutfeat = i; max_spread = spread; }
    }
  }
  const float split_val = (bbox[*cutfeat].low + bbox[*cutfeat].high) / 2;
  float mn, mx;
  kd_minmax(t, ind, count, *cutfeat, &mn, &mx);
  if (split_val < mn) *cutval = mn;
  else if (split_val > mx) *cutval = mx;
  else *cutval = split_val;
  int lim1, lim2;
  kd_plane_split(t, ind, count, *cutfeat, *cutval, &lim1, &lim2);
  if (lim1 > count / 2) *index = lim1;
  else if (lim2 < count / 2) *index = lim2;
  else *index = count / 2;
}
static int kd_divide(cfo_kdtree* t, int left, int right, kd_interval* bbox) { /* divideTree */
  const int me = t->nnodes++;
  if (right - left <= 15) {
    t->nodes[me].child1 = t->nodes[me].child2 = -1; t->nodes[me].left = left; t->nodes[me].right = right;
    for (int i = 0; i < 2; i++) bbox[i].low = bbox[i].high = t->pts[2 * t->vind[left] + i];
    for (int k = left + 1; k < right; k++)
      for (int i = 0; i < 2; i++) {
        const float v = t->pts[2 * t->vind[k] + i];
        if (bbox[i].low > v) bbox[i].low = v;
        if (bbox[i].high < v) bbox[i].high = v;
      }
  } else {
    int idx, cutfeat; float cutval;
    kd_middle_split(t, t->vind + left, right - left, &idx, &cutfeat, &cutval, bbox);
    t->nodes[me].divfeat = cutfeat;
    kd_interval lb[2] = {bbox[0], bbox[1]}, rb[2] = {bbox[0], bbox[1]};
    lb[cutfeat].high = cutval;
    const int c1 = kd_divide(t, left, left + idx, lb);
    rb[cutfeat].low = cutval;
    const int c2 = kd_divide(t, left + idx, right, rb);
    t->nodes[me].child1 = c1; t->nodes[me].child2 = c2;
    t->nodes[me].divlow = lb[cutfeat].high; t->nodes[me].divhigh = rb[cutfeat].low;
    for (int i = 0; i < 2; i++) {
      bbox[i].low = lb[i].low < rb[i].low ? lb[i].low : rb[i].low;
      bbox[i].high = lb[i].high > rb[i].high ? lb[i].high : rb[i].high;
    }
  }
  return me;
}
static cfo_kdtree* kd_build(const float* pts, int n) { /* buildIndex */
  cfo_kdtree* t = (cfo_kdtree*)calloc(1, sizeof(cfo_kdtree));
  t->n = n; t->pts = pts;
  t->vind = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  t->data = (float*)malloc(sizeof(float) * 2 * (size_t)(n > 0 ? n : 1));
  t->nodes = (kd_node*)malloc(sizeof(kd_node) * (size_t)(2 * n + 2));
  for (int i = 0; i < n; i++) t->vind[i] = i;
  if (n == 0) { t->root = -1; return t; }
  for (int i = 0; i < 2; i++) t->root_bbox[i].low = t->root_bbox[i].high = pts[i]; /* computeBoundingBox */
  for (int k = 1; k < n; k++)
    for (int i = 0; i < 2; i++) {
      if (pts[2 * k + i] < t->root_bbox[i].low) t->root_bbox[i].low = pts[2 * k + i];
      if (pts[2 * k + i] > t->root_bbox[i].high) t->root_bbox[i].high = pts[2 * k + i];
    }
  t->root = kd_divide(t, 0, n, t->root_bbox);
  for (int i = 0; i < n; i++) { t->data[2 * i] = pts[2 * t->vind[i]]; t->data[2 * i + 1] = pts[2 * t->vind[i] + 1]; }
  return t;
}
typedef struct { float worst; int index; } kd_result; /* KNNSimpleResultSet, capacity 1 */
static void kd_search_level(const cfo_kdtree* t, kd_result* rs, const float* vec, int node, float mindistsq, float* dists) { /* searchLevel, epsError = 1 */
  const kd_node* nd = &t->nodes[node];
  if (nd->child1 < 0) {
    const float worst_dist = rs->worst;
    for (int i = nd->left; i < nd->right; ++i) {
      float result = 0, diff; /* L2_Simple */
      diff = vec[0] - t->data[2 * i]; result += diff * diff;
      diff = vec[1] - t->data[2 * i + 1]; result += diff * diff;
      if (result < worst_dist) { /* addPoint: if (dist >= worst_distance_) return; */
        if (!(result >= rs->worst)) { rs->worst = result; rs->index = t->vind[i]; }
      }
    }
    return;
  }
  const int idx = nd->divfeat;
  const float val = vec[idx];
  const float diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
  int best, other; float cut_dist;
  if ((diff1 + diff2) < 0) { best = nd->child1; other = nd->child2; cut_dist = (val - nd->divhigh) * (val - nd->divhigh); }
  else { best = nd->child2; other = nd->child1; cut_dist = (val - nd->divlow) * (val - nd->divlow); }
  kd_search_level(t, rs, vec, best, mindistsq, dists);
  const float dst = dists[idx];
  mindistsq = mindistsq + cut_dist - dst;
  dists[idx] = cut_dist;
  if (mindistsq * 1.0f <= rs->worst) kd_search_level(t, rs, vec, other, mindistsq, dists);
  dists[idx] = dst;
}
static int kd_nearest(const cfo_kdtree* t, float qx, float qy, float* dist_out) { /* findNeighbors */
  if (t->n == 0) return -1;
  const float vec[2] = {qx, qy};
  float dists[2] = {0, 0}, distsq = 0; /* computeInitialDistances */
  for (int i = 0; i < 2; i++) {
    if (vec[i] < t->root_bbox[i].low) { dists[i] = (vec[i] - t->root_bbox[i].low) * (vec[i] - t->root_bbox[i].low); distsq += dists[i]; }
    if (vec[i] > t->root_bbox[i].high) { dists[i] = (vec[i] - t->root_bbox[i].high) * (vec[i] - t->root_bbox[i].high); distsq += dists[i]; }
  }
  kd_result rs = {FLT_MAX, -1};
  kd_search_level(t, &rs, vec, t->root, distsq, dists);
  *dist_out = rs.worst;
  return rs.index;
}

/* the restated FLANN search on its own (tests/test_oracle_sensitivity_cpu.py): 1-NN of nq query points among n 2-D float points */
void cfo_flann_nearest(const float* pts, int n, const float* queries, int nq, int* idx_out, float* dist_out) {
  cfo_kdtree* t = kd_build(pts, n);
  for (int i = 0; i < nq; i++) { float d = FLT_MAX; idx_out[i] = kd_nearest(t, queries[2 * i], queries[2 * i + 1], &d); dist_out[i] = d; }
  kd_free(t);
}

void cfo_scan_free(cfo_scan* s) {
  if (!s) return;
  free(s->pts); free(s->samples); free(s->cells); free(s->mean_f); free(s->gstart); free(s->gorder);
  kd_free(s->kd);
  free(s);
}
int cfo_scan_size(const cfo_scan* s) { return s ? s->ncells : 0; }
const cfo_cell* cfo_scan_cells(const cfo_scan* s) { return s->cells; }
int cfo_scan_num_samples(const cfo_scan* s) { return s->nsamp; }
const float* cfo_scan_samples(const cfo_scan* s) { return s->samples; }

/* GetClosestIdx (pointnormal.cpp:238-254): [3P] FLANN 1-NN over float (x,y), then d2 < d*d.
 * Exact-distance ties: lowest cell index (kd-tree visit order is not pinnable). */
int cfo_scan_closest(const cfo_scan* s, double px, double py, double d, int brute) {
  const float qx = (float)px, qy = (float)py;
  int best = -1;
  float bd = FLT_MAX;
  if (g_pert & CFO_PERT_NN_TIE_FLANN) { /* kd_cells.nearestKSearch(pnt, 1, ...) as FLANN does it */
    cfo_scan* sm = (cfo_scan*)s;
    if (!sm->kd) sm->kd = kd_build(s->mean_f, s->ncells);
    best = kd_nearest(sm->kd, qx, qy, &bd);
  } else if (brute || s->gw == 0) {
    for (int i = 0; i < s->ncells; i++) {
      const float dx = qx - s->mean_f[2 * i], dy = qy - s->mean_f[2 * i + 1];
      float d2 = dx * dx; d2 += dy * dy;
      if (d2 < bd || ((g_pert & CFO_PERT_NN_TIE_HIGH) && d2 == bd)) { bd = d2; best = i; }
    }
  } else {
    const double m = d * (1.0 + 1e-6) + 1e-6;
    int gx0 = (int)floor(((double)qx - m - s->gminx) / s->gcell), gx1 = (int)floor(((double)qx + m - s->gminx) / s->gcell);
    int gy0 = (int)floor(((double)qy - m - s->gminy) / s->gcell), gy1 = (int)floor(((double)qy + m - s->gminy) / s->gcell);
    if (gx0 < 0) gx0 = 0; if (gy0 < 0) gy0 = 0;
    if (gx1 >= s->gw) gx1 = s->gw - 1; if (gy1 >= s->gh) gy1 = s->gh - 1;
    for (int gy = gy0; gy <= gy1; gy++) {
      if (gx0 > gx1) break;
      const int a = s->gstart[gy * s->gw + gx0], b = s->gstart[gy * s->gw + gx1 + 1];
      for (int q = a; q < b; q++) {
        const int i = s->gorder[q];
        const float dx = qx - s->mean_f[2 * i], dy = qy - s->mean_f[2 * i + 1];
        float d2 = dx * dx; d2 += dy * dy;
        if (d2 < bd || (d2 == bd && ((g_pert & CFO_PERT_NN_TIE_HIGH) ? i > best : i < best))) { bd = d2; best = i; }
      }
    }
  }
  if (best >= 0 && (double)bd < d * d) return best;
  return -1;
}

/* ------------------------------------------------------------------------------------------
 * Stage 3: registration (n_scan_normal.cpp:82-187, 215-326, 344-452; registration.cpp:67-97)
 * ------------------------------------------------------------------------------------------ */

typedef struct { double l[4]; double t[2]; } aff2; /* linear row-major (00,01,10,11) + translation */

static aff2 aff_from_xyt(double x, double y, double th) { /* vectorToAffine3d, registration.cpp:130-136 */
  aff2 T; const double c = cos(th), s = sin(th);
  T.l[0] = c; T.l[1] = -s; T.l[2] = s; T.l[3] = c; T.t[0] = x; T.t[1] = y;
  return T;
}
static aff2 aff_mul(const aff2* A, const aff2* B) {
  aff2 C;
  C.l[0] = A->l[0] * B->l[0] + A->l[1] * B->l[2];
  C.l[1] = A->l[0] * B->l[1] + A->l[1] * B->l[3];
  C.l[2] = A->l[2] * B->l[0] + A->l[3] * B->l[2];
  C.l[3] = A->l[2] * B->l[1] + A->l[3] * B->l[3];
  C.t[0] = (A->l[0] * B->t[0] + A->l[1] * B->t[1]) + A->t[0];
  C.t[1] = (A->l[2] * B->t[0] + A->l[3] * B->t[1]) + A->t[1];
  return C;
}
static aff2 aff_inv(const aff2* A) { /* [3P] Eigen Affine inverse: general linear inverse, t' = -L^-1 t */
  aff2 I;
  const double det = A->l[0] * A->l[3] - A->l[1] * A->l[2];
  const double id = 1.0 / det;
  I.l[0] = A->l[3] * id; I.l[1] = -A->l[1] * id; I.l[2] = -A->l[2] * id; I.l[3] = A->l[0] * id;
  I.t[0] = -(I.l[0] * A->t[0] + I.l[1] * A->t[1]);
  I.t[1] = -(I.l[2] * A->t[0] + I.l[3] * A->t[1]);
  return I;
}
static void aff_to_xyt(const aff2* T, double v[3]) { /* Affine3dToVectorXYeZ, utils.cpp:115-122 */
  v[0] = T->t[0]; v[1] = T->t[1];
  v[2] = atan2(T->l[2], T->l[3]); /* [3P] eulerAngles(0,1,2)[2] of a pure yaw rotation */
}

typedef struct {
  double tm[2]; /* Ttar * tar_mean */
  double tn[2]; /* Ttar.linear() * tar_normal (P2L) */
  double L[3];  /* P2D sqrt information, lower triangle l00,l10,l11 */
  double s[2];  /* src mean (local) */
  double w;     /* weight after loss */
} match_t;

static double similarity(double x, double y) { return 2 * fmin(x, y) / (x + y); } /* registration.h:96 */
static double get_weight(int opt, double n1, double n2, double sim, double p1, double p2) { /* registration.cpp:67-76 */
  switch (opt) {
    case 0: return 1.0;
    case 1: return similarity(n1, n2);
    case 2: return sim;
    case 3: return similarity(p1, p2);
    case 4: return similarity(n1, n2) + sim + similarity(p1, p2);
    default: return 1.0;
  }
}

/* [3P] ceres::LossFunction::Evaluate for the losses of registration.cpp:78-97 (Ceres 2.0 forms) */
static void loss_eval(int loss, double a, double s, double rho[3]) {
  const double b = a * a;
  switch (loss) {
    case CFO_LOSS_HUBER:
      if (s > b) { const double r = sqrt(s); rho[0] = 2.0 * a * r - b; rho[1] = fmax(DBL_MIN, a / r); rho[2] = -rho[1] / (2.0 * s); }
      else { rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; }
      return;
    case CFO_LOSS_CAUCHY: {
      const double c = 1.0 / b, sum = 1.0 + s * c, inv = 1.0 / sum;
      rho[0] = b * log(sum); rho[1] = fmax(DBL_MIN, inv); rho[2] = -c * (inv * inv);
      return; }
    case CFO_LOSS_SOFTLONE: {
      const double c = 1.0 / b, sum = 1.0 + s * c, tmp = sqrt(sum);
      rho[0] = 2.0 * b * (tmp - 1.0); rho[1] = fmax(DBL_MIN, 1.0 / tmp); rho[2] = -(c * rho[1]) / (2.0 * sum);
      return; }
    case CFO_LOSS_TUKEY:
      if (s <= b) { const double v = 1.0 - s / b, v2 = v * v; rho[0] = b / 3.0 * (1.0 - v2 * v); rho[1] = v2; rho[2] = -2.0 / b * v; }
      else { rho[0] = b / 3.0; rho[1] = 0.0; rho[2] = 0.0; }
      return;
    case CFO_LOSS_COMBINED: { /* ComposedLoss(Huber(1), Cauchy(1)) = f(g(s)), registration.cpp:89-93 */
      double g[3], f[3];
      loss_eval(CFO_LOSS_CAUCHY, 1.0, s, g);
      loss_eval(CFO_LOSS_HUBER, 1.0, g[0], f);
      rho[0] = f[0]; rho[1] = f[1] * g[1]; rho[2] = f[2] * g[1] * g[1] + f[1] * g[2];
      return; }
    default: rho[0] = s; rho[1] = 1.0; rho[2] = 0.0; return; /* nullptr loss */
  }
}

/* rho(s), rho'(s), rho''(s) of the loss the registration builds for (loss, loss_limit) - registration.cpp:78-97 -> the Ceres 2.0 forms above; exported so
 * that tests can hold them against closed forms and, once it exists, against ceres::LossFunction::Evaluate itself (ref_golden.npz `ceres_loss_probe`) */
void cfo_loss_eval(int loss, double loss_limit, double s, double rho[3]) { loss_eval(loss, loss_limit, s, rho); }

typedef struct {
  const match_t* m; int nm; int cost; int loss; double loss_limit;
  /* soft constraint (n_scan_normal.cpp:373-377): residual L * alpha * (guess - x), no loss */
  int prior; double pL[9], pguess[3], palpha;
} problem_t;

/* [3P] ceres ResidualBlock::Evaluate + Corrector (rho'' <= 0 for every loss here => r~ = sqrt(rho')r).
 * cost = 1/2 sum rho(s); g = J~^T r~; H = J~^T J~ (00,01,02,11,12,22). Residual functors:
 * n_scan_normal.h:190-201 (P2L), :224-243 (P2D), :336-350 (P2P). */
static double evaluate_res(const problem_t* P, const double x[3], double g[3], double H[6], double* res_out);
static double evaluate(const problem_t* P, const double x[3], double g[3], double H[6]) { return evaluate_res(P, x, g, H, NULL); }
/* res_out (optional): the robustified residuals, sqrt(rho') r, in residual-block order ([3P] what
 * ceres::Problem::Evaluate returns with apply_loss_function = true) */
static double evaluate_res(const problem_t* P, const double x[3], double g[3], double H[6], double* res_out) {
  int nres_out = 0;
  const double c = cos(x[2]), s = sin(x[2]);
  double cost = 0;
  if (g) { g[0] = g[1] = g[2] = 0; for (int i = 0; i < 6; i++) H[i] = 0; }
  for (int i = 0; i < P->nm; i++) {
    const match_t* m = &P->m[i];
    const double px = (c * m->s[0] - s * m->s[1]) + x[0];
    const double py = (s * m->s[0] + c * m->s[1]) + x[1];
    const double dtx = -s * m->s[0] - c * m->s[1]; /* d p / d theta */
    const double dty = c * m->s[0] - s * m->s[1];
    double r[2], J[2][3];
    int nr;
    if (P->cost == CFO_COST_P2L) {
      nr = 1;
      r[0] = (px - m->tm[0]) * m->tn[0] + (py - m->tm[1]) * m->tn[1];
      J[0][0] = m->tn[0]; J[0][1] = m->tn[1]; J[0][2] = dtx * m->tn[0] + dty * m->tn[1];
    } else if (P->cost == CFO_COST_P2D) {
      nr = 2;
      const double dx = px - m->tm[0], dy = py - m->tm[1];
      r[0] = m->L[0] * dx; r[1] = m->L[1] * dx + m->L[2] * dy;
      J[0][0] = m->L[0]; J[0][1] = 0; J[0][2] = m->L[0] * dtx;
      J[1][0] = m->L[1]; J[1][1] = m->L[2]; J[1][2] = m->L[1] * dtx + m->L[2] * dty;
    } else {
      nr = 2;
      r[0] = m->tm[0] - px; r[1] = m->tm[1] - py;
      J[0][0] = -1; J[0][1] = 0; J[0][2] = -dtx;
      J[1][0] = 0; J[1][1] = -1; J[1][2] = -dty;
    }
    double sq = 0;
    for (int k = 0; k < nr; k++) sq += r[k] * r[k];
    double rho[3];
    loss_eval(P->loss, P->loss_limit, sq, rho);
    rho[0] *= m->w; rho[1] *= m->w; rho[2] *= m->w; /* ScaledLoss (n_scan_normal.cpp:277) */
    cost += 0.5 * rho[0];
    if (res_out) { const double sr = sqrt(rho[1]); for (int k = 0; k < nr; k++) res_out[nres_out++] = sr * r[k]; }
    if (g) {
      const double sr = sqrt(rho[1]);
      for (int k = 0; k < nr; k++) {
        const double rk = sr * r[k];
        const double j0 = sr * J[k][0], j1 = sr * J[k][1], j2 = sr * J[k][2];
        g[0] += j0 * rk; g[1] += j1 * rk; g[2] += j2 * rk;
        H[0] += j0 * j0; H[1] += j0 * j1; H[2] += j0 * j2;
        H[3] += j1 * j1; H[4] += j1 * j2; H[5] += j2 * j2;
      }
    }
  }
  if (P->prior) { /* mahalanobisDistanceError (n_scan_normal.h:259-290): r = L (alpha (guess - x)), J = -alpha L */
    double d[3], r[3];
    for (int k = 0; k < 3; k++) d[k] = P->palpha * (P->pguess[k] - x[k]);
    for (int i = 0; i < 3; i++) r[i] = P->pL[3 * i] * d[0] + P->pL[3 * i + 1] * d[1] + P->pL[3 * i + 2] * d[2];
    cost += 0.5 * (r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (res_out) for (int i = 0; i < 3; i++) res_out[nres_out++] = r[i];
    if (g) {
      for (int i = 0; i < 3; i++) {
        const double j0 = -P->palpha * P->pL[3 * i], j1 = -P->palpha * P->pL[3 * i + 1], j2 = -P->palpha * P->pL[3 * i + 2];
        g[0] += j0 * r[i]; g[1] += j1 * r[i]; g[2] += j2 * r[i];
        H[0] += j0 * j0; H[1] += j0 * j1; H[2] += j0 * j2;
        H[3] += j1 * j1; H[4] += j1 * j2; H[5] += j2 * j2;
      }
    }
  }
  return cost;
}

/* solve symmetric 3x3 A y = b by Cholesky; returns 0 if A is not positive definite */
static int chol3_solve(const double A[6], const double b[3], double y[3]) {
  const double a00 = A[0], a01 = A[1], a02 = A[2], a11 = A[3], a12 = A[4], a22 = A[5];
  if (!(a00 > 0)) return 0;
  const double l00 = sqrt(a00), l10 = a01 / l00, l20 = a02 / l00;
  const double d1 = a11 - l10 * l10;
  if (!(d1 > 0)) return 0;
  const double l11 = sqrt(d1), l21 = (a12 - l20 * l10) / l11;
  const double d2 = a22 - l20 * l20 - l21 * l21;
  if (!(d2 > 0)) return 0;
  const double l22 = sqrt(d2);
  const double z0 = b[0] / l00, z1 = (b[1] - l10 * z0) / l11, z2 = (b[2] - l20 * z0 - l21 * z1) / l22;
  y[2] = z2 / l22; y[1] = (z1 - l21 * y[2]) / l11; y[0] = (z0 - l10 * y[1] - l20 * y[2]) / l00;
  return isfinite(y[0]) && isfinite(y[1]) && isfinite(y[2]);
}

typedef struct {
  int num_iterations;     /* summary_.iterations.size() */
  int termination;        /* 0 CONVERGENCE 1 NO_CONVERGENCE 2 FAILURE */
  double final_cost;      /* min over pushed iterations (SetSummaryFinalCost) */
  double last_relative_decrease; /* iterations.back().relative_decrease */
} solve_summary;

/* [3P] ceres::Solve with default options + max_num_iterations (n_scan_normal.cpp:9,450):
 * trust-region Levenberg-Marquardt, Jacobi scaling, monotonic steps. */
static solve_summary lm_solve(const problem_t* P, double x[3], int max_iterations) {
  const double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
  const double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;
  const double max_radius = 1e16, min_radius = 1e-32;
  solve_summary S;
  double g[3], H[6];
  double x_cost = evaluate(P, x, g, H);
  double x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
  S.num_iterations = 1; S.final_cost = x_cost; S.last_relative_decrease = 0.0; S.termination = 1;
  double gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  if (gmax <= gradient_tolerance) { S.termination = 0; return S; }
  double scale[3];
  scale[0] = 1.0 / (1.0 + sqrt(H[0])); scale[1] = 1.0 / (1.0 + sqrt(H[3])); scale[2] = 1.0 / (1.0 + sqrt(H[5]));
  double radius = 1e4, decrease_factor = 2.0;
  int reuse_diagonal = 0, num_invalid = 0, iteration = 0;
  double diag[3] = {0, 0, 0};
  for (;;) {
    if (iteration >= max_iterations) { S.termination = 1; return S; }
    if (radius < min_radius) { S.termination = 0; return S; }
    iteration++;
    double Hs[6], gs[3];
    Hs[0] = H[0] * scale[0] * scale[0]; Hs[1] = H[1] * scale[0] * scale[1]; Hs[2] = H[2] * scale[0] * scale[2];
    Hs[3] = H[3] * scale[1] * scale[1]; Hs[4] = H[4] * scale[1] * scale[2]; Hs[5] = H[5] * scale[2] * scale[2];
    gs[0] = g[0] * scale[0]; gs[1] = g[1] * scale[1]; gs[2] = g[2] * scale[2];
    if (!reuse_diagonal) {
      diag[0] = fmin(fmax(Hs[0], min_lm_diagonal), max_lm_diagonal);
      diag[1] = fmin(fmax(Hs[3], min_lm_diagonal), max_lm_diagonal);
      diag[2] = fmin(fmax(Hs[5], min_lm_diagonal), max_lm_diagonal);
    }
    double lm[3];
    for (int i = 0; i < 3; i++) lm[i] = sqrt(diag[i] / radius);
    double Am[6] = {Hs[0] + lm[0] * lm[0], Hs[1], Hs[2], Hs[3] + lm[1] * lm[1], Hs[4], Hs[5] + lm[2] * lm[2]};
    double rhs[3] = {-gs[0], -gs[1], -gs[2]}, y[3];
    int valid = chol3_solve(Am, rhs, y);
    reuse_diagonal = 1;
    double model_cost_change = 0;
    if (valid) {
      const double Hy0 = Hs[0] * y[0] + Hs[1] * y[1] + Hs[2] * y[2];
      const double Hy1 = Hs[1] * y[0] + Hs[3] * y[1] + Hs[4] * y[2];
      const double Hy2 = Hs[2] * y[0] + Hs[4] * y[1] + Hs[5] * y[2];
      model_cost_change = -((y[0] * gs[0] + y[1] * gs[1] + y[2] * gs[2]) + 0.5 * (y[0] * Hy0 + y[1] * Hy1 + y[2] * Hy2));
      if (!(model_cost_change > 0.0)) valid = 0;
    }
    if (!valid) { /* HandleInvalidStep */
      if (++num_invalid >= 5) { S.termination = 2; return S; }
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1; /* StepIsInvalid = StepRejected(0) */
      S.num_iterations++; S.last_relative_decrease = 0.0;
      if (x_cost < S.final_cost) S.final_cost = x_cost;
      continue;
    }
    num_invalid = 0;
    double xc[3] = {x[0] + y[0] * scale[0], x[1] + y[1] * scale[1], x[2] + y[2] * scale[2]};
    const double cand_cost = evaluate(P, xc, NULL, NULL);
    const double d0 = x[0] - xc[0], d1 = x[1] - xc[1], d2 = x[2] - xc[2];
    const double step_norm = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    if (step_norm <= parameter_tolerance * (x_norm + parameter_tolerance)) { S.termination = 0; return S; }
    const double cost_change = x_cost - cand_cost;
    if (fabs(cost_change) <= function_tolerance * x_cost) { S.termination = 0; return S; }
    const double relative_decrease = cost_change / model_cost_change;
    S.num_iterations++;
    S.last_relative_decrease = relative_decrease;
    if (relative_decrease > min_relative_decrease) { /* HandleSuccessfulStep */
      x[0] = xc[0]; x[1] = xc[1]; x[2] = xc[2];
      x_norm = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
      x_cost = evaluate(P, x, g, H);
      { const double tq = 2.0 * relative_decrease - 1.0; /* [3P] Ceres: pow(2q-1, 3) */
        radius = radius / fmax(1.0 / 3.0, 1.0 - tq * tq * tq); }
      radius = fmin(max_radius, radius);
      decrease_factor = 2.0; reuse_diagonal = 0;
      if (x_cost < S.final_cost) S.final_cost = x_cost;
      gmax = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
      if (iteration >= max_iterations) { S.termination = 1; return S; }
      if (gmax <= gradient_tolerance) { S.termination = 0; return S; }
    } else { /* HandleUnsuccessfulStep */
      radius = radius / decrease_factor; decrease_factor *= 2.0; reuse_diagonal = 1;
      if (cand_cost < S.final_cost) S.final_cost = cand_cost;
    }
  }
}

/* AddScanPairCost for every (keyframe i -> current) pair (n_scan_normal.cpp:215-326, :359-367) */
static int build_problem(cfo_scan* const* scans, int n, double (*par)[3], const cfo_params* p, int itr,
                         int brute, match_t* M, int* num_residuals) {
  const double angle_outlier = cos(M_PI / 6.0);
  const double curr_radius = (itr == 1) ? 2 * p->assoc_radius : p->assoc_radius; /* :222 */
  const aff2 Tsrc = aff_from_xyt(par[n - 1][0], par[n - 1][1], par[n - 1][2]);
  const cfo_scan* src = scans[n - 1];
  int nm = 0;
  for (int i = 0; i + 1 < n; i++) {
    const cfo_scan* tar = scans[i];
    const aff2 Ttar = aff_from_xyt(par[i][0], par[i][1], par[i][2]);
    const aff2 Tinv = aff_inv(&Ttar);
    const aff2 T = aff_mul(&Tinv, &Tsrc); /* Tsrctotar, :224 */
    for (int j = 0; j < src->ncells; j++) {
      const cfo_cell* cs = &src->cells[j];
      const double qx = (T.l[0] * cs->mean[0] + T.l[1] * cs->mean[1]) + T.t[0];
      const double qy = (T.l[2] * cs->mean[0] + T.l[3] * cs->mean[1]) + T.t[1];
      const int ti = cfo_scan_closest(tar, qx, qy, curr_radius, brute);
      if (ti < 0) continue;
      const cfo_cell* ct = &tar->cells[ti];
      const double nx = T.l[0] * cs->normal[0] + T.l[1] * cs->normal[1];
      const double ny = T.l[2] * cs->normal[0] + T.l[3] * cs->normal[1];
      const double sim = fmax(nx * ct->normal[0] + ny * ct->normal[1], 0.0);
      if (!(sim > angle_outlier)) continue; /* :247 */
      match_t* m = &M[nm++];
      m->w = get_weight(p->weight_opt, (double)cs->nsamples, (double)ct->nsamples, sim, cs->scale, ct->scale);
      m->tm[0] = (Ttar.l[0] * ct->mean[0] + Ttar.l[1] * ct->mean[1]) + Ttar.t[0];
      m->tm[1] = (Ttar.l[2] * ct->mean[0] + Ttar.l[3] * ct->mean[1]) + Ttar.t[1];
      m->tn[0] = Ttar.l[0] * ct->normal[0] + Ttar.l[1] * ct->normal[1];
      m->tn[1] = Ttar.l[2] * ct->normal[0] + Ttar.l[3] * ct->normal[1];
      m->s[0] = cs->mean[0]; m->s[1] = cs->mean[1];
      m->L[0] = m->L[1] = m->L[2] = 0;
      if (p->cost == CFO_COST_P2D) { /* :290-299 */
        /* R*cov*R^T with the (symmetrised) cell covariance */
        const double a = ct->cov[0], b = ct->cov[1], c = ct->cov[2];
        const double r00 = Ttar.l[0], r01 = Ttar.l[1], r10 = Ttar.l[2], r11 = Ttar.l[3];
        const double m00 = r00 * a + r01 * b, m01 = r00 * b + r01 * c;
        const double m10 = r10 * a + r11 * b, m11 = r10 * b + r11 * c;
        const double c00 = (p->regularization + (m00 * r00 + m01 * r01)) * p->covar_scale;
        const double c10 = (0.0 + (m10 * r00 + m11 * r01)) * p->covar_scale;
        const double c01 = (0.0 + (m00 * r10 + m01 * r11)) * p->covar_scale;
        const double c11 = (p->regularization + (m10 * r10 + m11 * r11)) * p->covar_scale;
        const double det = c00 * c11 - c01 * c10, id = 1.0 / det; /* [3P] Eigen 2x2 inverse */
        const double i00 = c11 * id, i10 = -c10 * id, i11 = c00 * id;
        const double l00 = sqrt(i00), l10 = i10 / l00; /* [3P] Eigen LLT, lower */
        const double l11 = sqrt(i11 - l10 * l10);
        m->L[0] = l00; m->L[1] = l10; m->L[2] = l11;
      }
    }
  }
  *num_residuals = nm * (p->cost == CFO_COST_P2L ? 1 : 2);
  return nm;
}

static void default_cov(double* cov6) {
  for (int i = 0; i < 36; i++) cov6[i] = 0;
  cov6[0] = 0.1 * 0.1; cov6[7] = 0.1 * 0.1; cov6[35] = 0.01 * 0.01; /* n_scan_normal.cpp:173 */
}

/* n_scan_normal_reg::GetCost (n_scan_normal.cpp:188-213): associations and residual blocks for the given poses
 * (BuildOptimizationProblem with the object's current itr_, which only decides the association radius, :222),
 * then ceres::Problem::Evaluate with default options: score = 1/2 sum rho, residuals = robustified residuals.
 * Returns the number of residuals, or -1 where the reference returns false (<= 1 residuals). At most cap residuals
 * are written. */
int cfo_get_cost(cfo_scan* const* scans, int n, const double* poses_xyt, const cfo_params* p, int itr, int brute,
                 double* score, double* residuals, int cap) {
  if (n < 2 || n > 1024) return -1;
  double(*par)[3] = (double(*)[3])malloc(sizeof(double) * 3 * (size_t)n);
  for (int i = 0; i < n; i++) { /* Affine3dToVectorXYeZ (:196) */
    const aff2 T = aff_from_xyt(poses_xyt[3 * i], poses_xyt[3 * i + 1], poses_xyt[3 * i + 2]);
    aff_to_xyt(&T, par[i]);
  }
  const int nsrc = scans[n - 1]->ncells;
  match_t* M = (match_t*)malloc(sizeof(match_t) * (size_t)((n - 1) * (nsrc > 0 ? nsrc : 1)));
  problem_t P; P.m = M; P.cost = p->cost; P.loss = p->loss; P.loss_limit = p->loss_limit; P.prior = 0;
  int nres = 0;
  P.nm = build_problem(scans, n, par, p, itr, brute, M, &nres);
  int ret = -1;
  if (nres > 1) { /* :205-208 */
    double* all = (double*)malloc(sizeof(double) * (size_t)nres);
    const double c = evaluate_res(&P, par[n - 1], NULL, NULL, all);
    if (score) *score = c;
    if (residuals) memcpy(residuals, all, sizeof(double) * (size_t)(nres < cap ? nres : cap));
    free(all);
    ret = nres;
  }
  free(M); free(par);
  return ret;
}

/* ------------------------------------------------------------------------------------------
 * Cost-sampling covariance: OdometryKeyframeFuser::approximateCovarianceBySampling
 * (odometrykeyframefuser.cpp:261-380) with linspace (:497-524)
 * ------------------------------------------------------------------------------------------ */
static int linspace_d(double start, double end, int num, double* out) { /* :497-524 */
  if (num <= 0) return 0;
  if (num == 1) { out[0] = start; return 1; }
  const double delta = (end - start) / ((double)num - 1);
  for (int i = 0; i < num - 1; i++) out[i] = start + delta * i;
  out[num - 1] = end;
  return num;
}

/* cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 10): A = V diag(w) V^T */
static void jacobi_sym(int n, double* A, double* V, double* w) {
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0;
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
    if (off < 1e-300) break;
    for (int p = 0; p < n; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = A[p * n + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
        const double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        for (int k = 0; k < n; k++) {
          const double akp = A[k * n + p], akq = A[k * n + q];
          A[k * n + p] = c * akp - s * akq; A[k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; k++) {
          const double apk = A[p * n + k], aqk = A[q * n + k];
          A[p * n + k] = c * apk - s * aqk; A[q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; k++) {
          const double vkp = V[k * n + p], vkq = V[k * n + q];
          V[k * n + p] = c * vkp - s * vkq; V[k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

/* minimum-norm least-squares solution of A c = b (A: m x 10), [3P] what Eigen's bdcSvd().solve() returns. The columns differ
 * by six orders of magnitude (yaw^2 ~ 5e-6 next to 1), so they are scaled to unit norm D before the normal matrix of A D is
 * eigen-decomposed and pseudo-inverted: y0 = (A D)^+ b, c0 = D y0 is A least-squares solution. When A is rank deficient
 * (samples_per_axis = 2: x^2, y^2, yaw^2 and 1 are parallel columns) the minimum-norm one is c0 + B z with B = D N, N the null
 * space of A D, and z from the small problem min |c0 + B z| (normal equations B^T B z = -B^T c0, Gaussian elimination). */
static void lstsq10(int m, const double* A, const double* b, double c[10]) {
  double scale[10], N[100], V[100], w[10], rhs[10];
  for (int j = 0; j < 10; j++) {
    double s = 0;
    for (int i = 0; i < m; i++) s += A[i * 10 + j] * A[i * 10 + j];
    scale[j] = s > 0 ? 1.0 / sqrt(s) : 0.0;
  }
  for (int j = 0; j < 10; j++) {
    for (int k = 0; k < 10; k++) {
      double s = 0;
      for (int i = 0; i < m; i++) s += A[i * 10 + j] * A[i * 10 + k];
      N[j * 10 + k] = s * scale[j] * scale[k];
    }
    double s = 0;
    for (int i = 0; i < m; i++) s += A[i * 10 + j] * b[i];
    rhs[j] = s * scale[j];
  }
  jacobi_sym(10, N, V, w);
  double wmax = 0;
  for (int j = 0; j < 10; j++) if (w[j] > wmax) wmax = w[j];
  double y[10];
  int nullcol[10], p = 0;
  for (int j = 0; j < 10; j++) {
    double s = 0;
    for (int k = 0; k < 10; k++) s += V[k * 10 + j] * rhs[k];
    if (w[j] > 1e-12 * wmax) y[j] = s / w[j];
    else { y[j] = 0.0; nullcol[p++] = j; }
  }
  for (int k = 0; k < 10; k++) {
    double s = 0;
    for (int j = 0; j < 10; j++) s += V[k * 10 + j] * y[j];
    c[k] = s * scale[k];
  }
  if (p > 0) { /* rank deficient: move along the null space to the shortest solution */
    double B[100], M[100], g[10], z[10];
    for (int k = 0; k < 10; k++) for (int a = 0; a < p; a++) B[k * 10 + a] = scale[k] * V[k * 10 + nullcol[a]];
    for (int a = 0; a < p; a++) {
      for (int q = 0; q < p; q++) { double s = 0; for (int k = 0; k < 10; k++) s += B[k * 10 + a] * B[k * 10 + q]; M[a * 10 + q] = s; }
      double s = 0; for (int k = 0; k < 10; k++) s += B[k * 10 + a] * c[k];
      g[a] = -s;
    }
    for (int a = 0; a < p; a++) { /* elimination with partial pivoting; a null direction made of all-zero columns drops out */
      int piv = a;
      for (int r = a + 1; r < p; r++) if (fabs(M[r * 10 + a]) > fabs(M[piv * 10 + a])) piv = r;
      if (piv != a) { for (int q = 0; q < p; q++) { const double t = M[a * 10 + q]; M[a * 10 + q] = M[piv * 10 + q]; M[piv * 10 + q] = t; } const double t = g[a]; g[a] = g[piv]; g[piv] = t; }
      if (fabs(M[a * 10 + a]) < 1e-300) continue;
      for (int r = a + 1; r < p; r++) {
        const double f = M[r * 10 + a] / M[a * 10 + a];
        for (int q = a; q < p; q++) M[r * 10 + q] -= f * M[a * 10 + q];
        g[r] -= f * g[a];
      }
    }
    for (int a = p - 1; a >= 0; a--) {
      double s = g[a];
      for (int q = a + 1; q < p; q++) s -= M[a * 10 + q] * z[q];
      z[a] = fabs(M[a * 10 + a]) < 1e-300 ? 0.0 : s / M[a * 10 + a];
    }
    for (int k = 0; k < 10; k++) { double s = 0; for (int a = 0; a < p; a++) s += B[k * 10 + a] * z[a]; c[k] += s; }
  }
}

/* Returns 1 and fills cov6 (36 doubles, row-major) where the reference returns true. sample_costs (optional, steps^3
 * doubles) receives the sampled costs in the reference's loop order (theta outer, x, y inner). final_cost and
 * num_residuals are those of the preceding Register (GetCovarianceScaler, n_scan_normal.cpp:435-441). */
int cfo_cov_by_sampling(cfo_scan* const* scans, int n, const double* poses_xyt, const cfo_params* p, int itr, int brute,
                        double xy_range, double yaw_range, int steps, double cov_scaler, double final_cost, int num_residuals,
                        double* cov6, double* sample_costs) {
  if (steps < 1 || steps > 64 || n < 2) return 0;
  const int m = steps * steps * steps;
  double* xs = (double*)malloc(sizeof(double) * (size_t)steps);
  double* ths = (double*)malloc(sizeof(double) * (size_t)steps);
  linspace_d(-xy_range * 0.5, xy_range * 0.5, steps, xs);   /* :277-289 */
  linspace_d(-yaw_range * 0.5, yaw_range * 0.5, steps, ths);
  double* A = (double*)malloc(sizeof(double) * 10 * (size_t)m);
  double* b = (double*)malloc(sizeof(double) * (size_t)m);
  double* poses = (double*)malloc(sizeof(double) * 3 * (size_t)n);
  memcpy(poses, poses_xyt, sizeof(double) * 3 * (size_t)n);
  const int L = 3 * (n - 1);
  double sample_cost = 0; /* not reset when GetCost fails: the reference ignores its return value (:305) */
  int k = 0;
  for (int it = 0; it < steps; it++)
    for (int ix = 0; ix < steps; ix++)
      for (int iy = 0; iy < steps; iy++) {
        poses[L] = xs[ix] + poses_xyt[L]; poses[L + 1] = xs[iy] + poses_xyt[L + 1];
        poses[L + 2] = ths[it] + poses_xyt[L + 2]; /* Rz(theta_s) * Rz(yaw) */
        double sc = 0;
        if (cfo_get_cost(scans, n, poses, p, itr, brute, &sc, NULL, 0) >= 0) sample_cost = sc;
        const double x = xs[ix], y = xs[iy], z = ths[it];
        double* r = A + 10 * (size_t)k;
        r[0] = x * x; r[1] = y * y; r[2] = z * z; r[3] = x * y; r[4] = y * z; r[5] = z * x; r[6] = x; r[7] = y; r[8] = z; r[9] = 1.0;
        b[k] = sample_cost;
        if (sample_costs) sample_costs[k] = sample_cost;
        k++;
      }
  double c[10];
  lstsq10(m, A, b, c);
  free(xs); free(ths); free(A); free(b); free(poses);
  double H[9] = {2 * c[0], c[3], c[5], c[3], 2 * c[1], c[4], c[5], c[4], 2 * c[2]}; /* :340-343 */
  double Hc[9], V[9], w[3];
  memcpy(Hc, H, sizeof(H));
  jacobi_sym(3, Hc, V, w);
  if (!(w[0] > 0.0 && w[1] > 0.0 && w[2] > 0.0)) return 0; /* not convex (:355-358) */
  if (num_residuals - 3 == 0) return 0;                     /* GetCovarianceScaler false */
  const double score_scale = final_cost / (double)(num_residuals - 3);
  double C3[9]; /* 2 * H^-1 * score_scale * scaler, H^-1 = V diag(1/w) V^T */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int q = 0; q < 3; q++) s += V[i * 3 + q] * V[j * 3 + q] / w[q];
      C3[i * 3 + j] = 2.0 * s * score_scale * cov_scaler;
    }
  for (int i = 0; i < 36; i++) cov6[i] = (i % 7 == 0) ? 1.0 : 0.0; /* :367-374 */
  cov6[0] = C3[0]; cov6[1] = C3[1]; cov6[6] = C3[3]; cov6[7] = C3[4];
  cov6[35] = C3[8]; cov6[5] = C3[2]; cov6[11] = C3[5]; cov6[30] = C3[6]; cov6[31] = C3[7];
  return 1;
}

static int register_impl(cfo_scan* const* scans, int n, double* poses_xyt, double* cov6, const cfo_params* p,
                         int brute, cfo_reg_summary* out, const double* prior_cov6) {
  cfo_reg_summary S;
  memset(&S, 0, sizeof(S));
  if (n < 2 || n > 1024) { if (out) *out = S; return 0; }
  double(*par)[3] = (double(*)[3])malloc(sizeof(double) * 3 * (size_t)n);
  for (int i = 0; i < n; i++) { /* Affine3dToVectorXYeZ of vectorToAffine3d: theta -> atan2(sin,cos) */
    const aff2 T = aff_from_xyt(poses_xyt[3 * i], poses_xyt[3 * i + 1], poses_xyt[3 * i + 2]);
    aff_to_xyt(&T, par[i]);
  }
  const int nsrc = scans[n - 1]->ncells;
  match_t* M = (match_t*)malloc(sizeof(match_t) * (size_t)((n - 1) * (nsrc > 0 ? nsrc : 1)));
  double tsrc_last[3] = {poses_xyt[3 * (n - 1)], poses_xyt[3 * (n - 1) + 1], poses_xyt[3 * (n - 1) + 2]};
  int success = 1;
  double prev_par[3] = {par[n - 1][0], par[n - 1][1], par[n - 1][2]};
  double prev_score = DBL_MAX;
  problem_t P; P.m = M; P.nm = 0; P.cost = p->cost; P.loss = p->loss; P.loss_limit = p->loss_limit; P.prior = 0;
  int prior_ok = 0;
  if (prior_cov6) { /* :373-376: guess_inf_sqrt = Cov6to3(cov).inverse().llt().matrixL(); alpha = sqrt(N_src) */
    const double* C = prior_cov6;
    const double a = C[0], b = C[1], c = C[5], d = C[6], e = C[7], f5 = C[11], g6 = C[30], h = C[31], i9 = C[35]; /* Cov6to3 (registration.cpp:123-129) */
    const double A00 = e * i9 - f5 * h, A01 = c * h - b * i9, A02 = b * f5 - c * e;
    const double A10 = f5 * g6 - d * i9, A11 = a * i9 - c * g6, A12 = c * d - a * f5;
    const double A20 = d * h - e * g6, A21 = b * g6 - a * h, A22 = a * e - b * d;
    const double det = a * A00 + b * A10 + c * A20;
    const double I[9] = {A00 / det, A01 / det, A02 / det, A10 / det, A11 / det, A12 / det, A20 / det, A21 / det, A22 / det};
    /* lower Cholesky factor of the (symmetric) information matrix, reading its lower triangle like Eigen's LLT */
    const double l00 = sqrt(I[0]), l10 = I[3] / l00, l20 = I[6] / l00;
    const double l11 = sqrt(I[4] - l10 * l10), l21 = (I[7] - l20 * l10) / l11;
    const double l22 = sqrt(I[8] - l20 * l20 - l21 * l21);
    const double L[9] = {l00, 0, 0, l10, l11, 0, l20, l21, l22};
    memcpy(P.pL, L, sizeof(L));
    P.pguess[0] = par[n - 1][0]; P.pguess[1] = par[n - 1][1]; P.pguess[2] = par[n - 1][2]; /* Affine3dToEigVectorXYeZ(Tsrc.back()) (:93-94) */
    P.palpha = sqrt((double)scans[n - 1]->ncells);
    prior_ok = 1;
  }
  int nres = 0;
  solve_summary ss; memset(&ss, 0, sizeof(ss));
  int itr;
  for (itr = 1; itr <= p->max_itr_association && success; itr++) { /* :102 */
    P.nm = build_problem(scans, n, par, p, itr, brute, M, &nres);
    if (nres <= 1) { success = 0; break; } /* :370-371, :114-115 */
    if (prior_ok) { P.prior = 1; nres += 3; } /* the prior block joins after the residual-count check (:370-377) */
    ss = lm_solve(&P, par[n - 1], p->max_solver_iterations);
    success = (ss.termination != 2); /* IsSolutionUsable */
    if (success) { tsrc_last[0] = par[n - 1][0]; tsrc_last[1] = par[n - 1][1]; tsrc_last[2] = par[n - 1][2]; } /* :119-121 */
    if (itr - 1 < CFO_MAX_OUTER) {
      S.inner_iterations[itr - 1] = ss.num_iterations; S.termination[itr - 1] = ss.termination;
      S.outer_cost[itr - 1] = ss.final_cost;
      S.outer_pose[itr - 1][0] = par[n - 1][0]; S.outer_pose[itr - 1][1] = par[n - 1][1]; S.outer_pose[itr - 1][2] = par[n - 1][2];
    }
    const double current_score = ss.final_cost;
    const double rel_improvement = (prev_score - current_score) / prev_score;
    if (itr > p->min_itr) { /* :134-149 */
      if (prev_score < current_score) { par[n - 1][0] = prev_par[0]; par[n - 1][1] = prev_par[1]; par[n - 1][2] = prev_par[2]; break; }
      else if (rel_improvement < 0.00001) break;
      else if (ss.last_relative_decrease < 0.00001 || ss.num_iterations == 1) break;
    }
    prev_score = current_score;
    prev_par[0] = par[n - 1][0]; prev_par[1] = par[n - 1][1]; prev_par[2] = par[n - 1][2];
  }
  S.outer_iterations = itr;
  S.usable = success;
  S.num_residuals = nres; S.num_residual_blocks = P.nm; S.final_cost = ss.final_cost;
  int ret = 0;
  if (success) {
    S.score = ss.final_cost / nres; /* :166 */
    if (cov6) default_cov(cov6);
    /* Tsrc[i] = vectorToAffine3d(parameters[i]) -> the caller reads (x,y,atan2(sin,cos)) */
    for (int i = 0; i < n; i++) { poses_xyt[3 * i] = par[i][0]; poses_xyt[3 * i + 1] = par[i][1]; poses_xyt[3 * i + 2] = par[i][2]; }
    /* GetCovariance (:392-433): [3P] ceres::Covariance = (J~^T J~)^-1 at the final parameters */
    double g[3], H[6];
    evaluate(&P, par[n - 1], g, H);
    const double a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5];
    const double C00 = d * f - e * e, C01 = c * e - b * f, C02 = b * e - c * d;
    const double det = a * C00 + b * C01 + c * C02;
    const int dof = nres - 3;
    if (det > 0 && isfinite(det) && dof != 0) {
      const double sc = 30 * (ss.final_cost / dof) / det;
      const double c00 = sc * C00, c01 = sc * C01, c02 = sc * C02, c11 = sc * (a * f - c * c), c22 = sc * (a * d - b * b);
      if (cov6) {
        for (int i = 0; i < 36; i++) cov6[i] = (i % 7 == 0) ? 1.0 : 0.0;
        cov6[0] = c00; cov6[1] = c01; cov6[6] = c01; cov6[7] = c11;
        cov6[35] = c22; cov6[5] = c02; cov6[30] = c02; /* (1,5)/(5,1) left 0: :426-430 */
      }
      ret = 1;
    }
  } else {
    poses_xyt[3 * (n - 1)] = tsrc_last[0]; poses_xyt[3 * (n - 1) + 1] = tsrc_last[1]; poses_xyt[3 * (n - 1) + 2] = tsrc_last[2];
  }
  S.success = ret;
  if (out) *out = S;
  free(M); free(par);
  return ret;
}

int cfo_register(cfo_scan* const* scans, int n, double* poses_xyt, double* cov6, const cfo_params* p,
                 int brute, cfo_reg_summary* out) {
  return register_impl(scans, n, poses_xyt, cov6, p, brute, out, NULL);
}

/* Register(..., soft_constraints = true): prior_cov6 = reg_cov.back() as passed in (36 doubles, row-major) */
int cfo_register_soft(cfo_scan* const* scans, int n, double* poses_xyt, const double* prior_cov6, double* cov6,
                      const cfo_params* p, int brute, cfo_reg_summary* out) {
  return register_impl(scans, n, poses_xyt, cov6, p, brute, out, prior_cov6);
}

/* ------------------------------------------------------------------------------------------
 * Caller: OdometryKeyframeFuser::processFrame (odometrykeyframefuser.cpp:143-259)
 * ------------------------------------------------------------------------------------------ */
#define CFO_MAX_KEYFRAMES 64
struct cfo_fuser {
  cfo_params par;
  aff2 T_prev, Tmot, Tcurrent;
  cfo_scan* kf_scan[CFO_MAX_KEYFRAMES];
  aff2 kf_pose[CFO_MAX_KEYFRAMES];
  int nkf;
  cfo_scan* last_scan; int last_scan_owned;
  cfo_reg_summary last;
  double timers[4];
  long frames;
  /* cov_current (odometrykeyframefuser.h:204) and the cost-sampling option (Parameters::estimate_cov_by_sampling and friends, :104-110) */
  double cov_current[36];
  int cov_sampling, cov_steps; double cov_xy_range, cov_yaw_range, cov_scaler;
};

static aff2 aff_identity(void) { aff2 T; T.l[0] = 1; T.l[1] = 0; T.l[2] = 0; T.l[3] = 1; T.t[0] = T.t[1] = 0; return T; }

cfo_fuser* cfo_fuser_create(const cfo_params* p) {
  cfo_fuser* f = (cfo_fuser*)calloc(1, sizeof(cfo_fuser));
  f->par = *p;
  if (f->par.submap_scan_size > CFO_MAX_KEYFRAMES) f->par.submap_scan_size = CFO_MAX_KEYFRAMES;
  f->T_prev = f->Tmot = f->Tcurrent = aff_identity();
  return f;
}
void cfo_fuser_free(cfo_fuser* f) {
  if (!f) return;
  for (int i = 0; i < f->nkf; i++) cfo_scan_free(f->kf_scan[i]);
  if (f->last_scan_owned) cfo_scan_free(f->last_scan);
  free(f);
}
int cfo_fuser_num_keyframes(const cfo_fuser* f) { return f->nkf; }
/* par.estimate_cov_by_sampling, cov_sampling_xy_range, cov_sampling_yaw_range, cov_sampling_samples_per_axis, cov_sampling_covariance_scaler */
void cfo_fuser_set_cov_sampling(cfo_fuser* f, int enable, double xy_range, double yaw_range, int steps, double scaler) {
  f->cov_sampling = enable; f->cov_xy_range = xy_range; f->cov_yaw_range = yaw_range; f->cov_steps = steps; f->cov_scaler = scaler;
}
/* cov_current as pointcloudCallback(..., Covariance& cov_curr) hands it back (odometrykeyframefuser.cpp:397-411): 36 doubles row-major */
void cfo_fuser_last_cov(const cfo_fuser* f, double cov6[36]) { memcpy(cov6, f->cov_current, sizeof(double) * 36); }
const cfo_reg_summary* cfo_fuser_last_summary(const cfo_fuser* f) { return &f->last; }
const cfo_scan* cfo_fuser_last_scan(const cfo_fuser* f) { return f->last_scan; }
void cfo_fuser_timers(const cfo_fuser* f, double t[4]) { for (int i = 0; i < 4; i++) t[i] = f->timers[i]; }

static void add_to_reference(cfo_fuser* f, cfo_scan* s, const aff2* T) { /* :470-476 */
  f->kf_scan[f->nkf] = s; f->kf_pose[f->nkf] = *T; f->nkf++;
  if (f->nkf > f->par.submap_scan_size) {
    cfo_scan_free(f->kf_scan[0]);
    for (int i = 0; i + 1 < f->nkf; i++) { f->kf_scan[i] = f->kf_scan[i + 1]; f->kf_pose[i] = f->kf_pose[i + 1]; }
    f->nkf--;
  }
}

int cfo_fuser_process_cloud(cfo_fuser* f, float* xyi, int n, double pose_xyt[3]) {
  const cfo_params* p = &f->par;
  double t0 = now_s();
  const aff2 TprevMot = f->Tmot; /* :146 */
  if (p->compensate) { double mot[3]; aff_to_xyt(&TprevMot, mot); cfo_compensate(xyi, n, mot, p->radar_ccw); }
  double t1 = now_s();
  if (f->last_scan_owned) { cfo_scan_free(f->last_scan); f->last_scan_owned = 0; }
  cfo_scan* cur = cfo_scan_create(xyi, n, p, 0); /* :161 */
  f->last_scan = cur;
  double t2 = now_s();
  f->timers[1] += t1 - t0; f->timers[2] += t2 - t1;
  if (!cur) return -1;
  const aff2 Tguess = aff_mul(&f->T_prev, &TprevMot); /* :166, use_guess forced true (offline_odometry.cpp:273) */
  f->frames++;
  if (f->nkf == 0) { /* :171-177 */
    const aff2 I = aff_identity();
    add_to_reference(f, cur, &I);
    memset(&f->last, 0, sizeof(f->last));
    aff_to_xyt(&f->Tcurrent, pose_xyt);
    return 0;
  }
  /* FormatScans (:478-494): keyframes oldest first, current last */
  cfo_scan* scans[CFO_MAX_KEYFRAMES + 1];
  double poses[3 * (CFO_MAX_KEYFRAMES + 1)], cov6[36];
  const int ns = f->nkf + 1;
  for (int i = 0; i < f->nkf; i++) { scans[i] = f->kf_scan[i]; aff_to_xyt(&f->kf_pose[i], &poses[3 * i]); }
  scans[ns - 1] = cur; aff_to_xyt(&Tguess, &poses[3 * (ns - 1)]);
  for (int i = 0; i < 36; i++) cov6[i] = (i % 7 == 0) ? 1.0 : 0.0; /* FormatScans: cov_vek entries are Identity66 (:486-490) */
  cfo_register(scans, ns, poses, cov6, p, 0, &f->last); /* result ignored: shadowed 'success' (:184-186) */
  memcpy(f->cov_current, cov6, sizeof(cov6)); /* :196 cov_current = cov_vek.back() */
  if (f->cov_sampling) { /* :202-208; the samples are GetCost calls of radar_reg, whose itr_ is what Register left */
    double cs[36];
    if (cfo_cov_by_sampling(scans, ns, poses, p, f->last.outer_iterations, 0, f->cov_xy_range, f->cov_yaw_range, f->cov_steps, f->cov_scaler,
                            f->last.final_cost, f->last.num_residuals, cs, NULL))
      memcpy(f->cov_current, cs, sizeof(cs));
  }
  double t3 = now_s();
  f->timers[3] += t3 - t2;
  aff2 Tcurrent = aff_from_xyt(poses[3 * (ns - 1)], poses[3 * (ns - 1) + 1], poses[3 * (ns - 1) + 2]); /* :195 */
  const aff2 Tpi = aff_inv(&f->T_prev);
  const aff2 Tmot_current = aff_mul(&Tpi, &Tcurrent);
  { /* AccelerationVelocitySanityCheck (:76-94) */
    const double dt = 0.25, lim = 200;
    const double vel = sqrt(Tmot_current.t[0] * Tmot_current.t[0] + Tmot_current.t[1] * Tmot_current.t[1]) / dt;
    const double ax = (Tmot_current.t[0] - f->Tmot.t[0]) / (dt * dt), ay = (Tmot_current.t[1] - f->Tmot.t[1]) / (dt * dt);
    const double acc = sqrt(ax * ax + ay * ay);
    if (acc > lim || vel > lim) Tcurrent = Tguess; /* :198-199 */
  }
  f->Tmot = aff_mul(&Tpi, &Tcurrent); /* :200 */
  f->Tcurrent = Tcurrent;
  /* KeyFrameBasedFuse (:62-73) */
  const aff2 Tki = aff_inv(&f->kf_pose[f->nkf - 1]);
  const aff2 Tkeydiff = aff_mul(&Tki, &Tcurrent);
  int fuse = 1;
  if (p->use_keyframe) {
    const double tn = sqrt(Tkeydiff.t[0] * Tkeydiff.t[0] + Tkeydiff.t[1] * Tkeydiff.t[1]);
    const double rot = fabs(atan2(Tkeydiff.l[2], Tkeydiff.l[3]));
    fuse = (tn > p->min_keyframe_dist) || (rot > p->min_keyframe_rot_deg * M_PI / 180.0);
  }
  if (fuse) add_to_reference(f, cur, &Tcurrent); /* :234-247 */
  else f->last_scan_owned = 1;
  f->T_prev = Tcurrent; /* :257 */
  aff_to_xyt(&Tcurrent, pose_xyt);
  return 0;
}

int cfo_fuser_process_polar(cfo_fuser* f, const uint8_t* img, int A, int R, double pose_xyt[3]) {
  const cfo_params* p = &f->par;
  const int k = p->k_strongest;
  double t0 = now_s();
  uint32_t* slots = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)A * (size_t)k);
  float* xyi = (float*)malloc(sizeof(float) * 3 * (size_t)A * (size_t)k);
  float* xyi_peaks = (float*)malloc(sizeof(float) * 3 * (size_t)A * (size_t)k);
  cfo_filter(img, A, R, (int)p->z_min, k, slots); /* radar_driver.cpp:58: float z_min -> const int */
  const int n = cfo_cloud(slots, A, k, p->range_res, p->min_distance, 0, xyi);       /* :59 */
  const int np = cfo_cloud(slots, A, k, p->range_res, p->min_distance, 1, xyi_peaks); /* :60 */
  f->timers[0] += now_s() - t0;
  if (p->compensate) { double mot[3]; aff_to_xyt(&f->Tmot, mot); cfo_compensate(xyi_peaks, np, mot, p->radar_ccw); } /* odometrykeyframefuser.cpp:149 */
  const int rc = cfo_fuser_process_cloud(f, xyi, n, pose_xyt);
  free(slots); free(xyi); free(xyi_peaks);
  return rc;
}
