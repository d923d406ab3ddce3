/*
 * cfear_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, NOT THE PRODUCT).
 *
 * Single-thread C99 restatement of the CFEAR per-scan hot path of
 * dan11003/CFEAR_Radarodometry_code_public, used ONLY by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg to check / time
 * beside the HIP path.  The product library (libcfear_hip.so) never links,
 * loads or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for
 * this path and cannot be compiled here (it needs ROS1, PCL/FLANN, Ceres,
 * Eigen3, OpenCV, Boost -- none are in the image).  The arithmetic of those
 * third-party libraries is restated from their published algorithms (marked
 * [3P] in cfear_oracle.c) and anchored on the reference's own call sites.
 *
 * All file:line citations are relative to /root/reference.
 */
#ifndef CFEAR_ORACLE_H
#define CFEAR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* enums mirror include/cfear_radarodometry/registration.h:48-60 */
enum { CFO_COST_P2P = 0, CFO_COST_P2L = 1, CFO_COST_P2D = 2 };
enum { CFO_LOSS_NONE = 0, CFO_LOSS_HUBER = 1, CFO_LOSS_CAUCHY = 2, CFO_LOSS_SOFTLONE = 3,
       CFO_LOSS_COMBINED = 4, CFO_LOSS_TUKEY = 5 };

/* Same field order/layout as cfear_params in include/cfear_hip.h (tests assert the sizes match). */
typedef struct cfo_params {
  /* radarDriver::Parameters (radar_driver.h:40-45): floats on purpose */
  float z_min;
  float range_res;
  float min_distance;
  int32_t k_strongest;
  /* OdometryKeyframeFuser::Parameters (odometrykeyframefuser.h:86-102) */
  double res;               /* radius r; narrowed to float at pointnormal.h:118 */
  double downsample_factor; /* MapPointNormal::downsample_factor, pointnormal.cpp:5 */
  int32_t weight_intensity;
  int32_t cost;             /* CFO_COST_* */
  int32_t loss;             /* CFO_LOSS_* */
  int32_t weight_opt;       /* registration.h:50 */
  double loss_limit;
  double covar_scale;       /* SetD2dPar, n_scan_normal.h:53 */
  double regularization;
  int32_t submap_scan_size;
  int32_t compensate;
  int32_t radar_ccw;
  int32_t use_keyframe;
  double min_keyframe_dist;
  double min_keyframe_rot_deg;
  /* n_scan_normal.h:75, n_scan_normal.cpp:9, registration.h:122 */
  int32_t max_itr_association; /* 8 */
  int32_t min_itr;             /* 3 */
  int32_t max_solver_iterations; /* 20 */
  int32_t reserved0;
  double assoc_radius;         /* 2.0 */
} cfo_params;

void cfo_default_params(cfo_params* p);

/* ---- [3P] sensitivity modes (tests/run_3p_sensitivity.py; process-wide, 0 = the oracle as specified). Each bit swaps one piece
 * of third-party behaviour that the reference's sources do not pin for another admissible one, so that its effect on poses and
 * iteration counts over long drives can be BOUNDED (profiles/r04_3p_sensitivity.json):
 *   VOXEL_REVERSE / VOXEL_RANDOM / VOXEL_STDSORT  order of the points inside a voxel when the float centroid is summed
 *       (PCL VoxelGrid sorts on the voxel index with an unstable sort; STDSORT = libstdc++ std::sort on the index only, i.e.
 *       exactly PCL <= 1.9 on Ubuntu: needs cfo_set_voxel_sorter with the function of oracle/stdsort_perm.cpp);
 *   SUM_REVERSE / SUM_PAIRWISE  order in which a cell's weights, mean and covariance terms are added (pointnormal.cpp:13-33);
 *   WSUM_EIGEN_REDUX            only `w.sum()` (pointnormal.cpp:19) in the order of Eigen's vectorised redux (SSE2 packets);
 *   EIG_JACOBI                  the 2x2 eigen-decomposition by a Jacobi rotation instead of the closed form (:39-45);
 *   NN_TIE_HIGH                 exact-distance ties of the 1-NN search go to the highest cell index instead of the lowest;
 *   NN_TIE_FLANN   [3P-recall]  the 1-NN search of GetClosestIdx (pointnormal.cpp:238-254) by a restatement of what it calls:
 *       pcl::KdTreeFLANN<PointXY>::nearestKSearch -> flann::KDTreeSingleIndex (L2_Simple<float>, leaf_max_size 15, reorder, eps 0,
 *       KNNSimpleResultSet of one): bounding-box middle split with the planeSplit partition, best-child-first descent, a point is
 *       taken only if STRICTLY nearer than the best so far - so among exactly equidistant cells the first one the descent visits wins,
 *       which is a property of the tree's layout, not of the cell index. Always a true nearest neighbour; equal to the brute-force
 *       answer whenever the minimum is unique (tests/test_oracle_sensitivity_cpu.py). */
enum { CFO_PERT_VOXEL_REVERSE = 1, CFO_PERT_VOXEL_RANDOM = 2, CFO_PERT_VOXEL_STDSORT = 4, CFO_PERT_SUM_REVERSE = 8,
       CFO_PERT_SUM_PAIRWISE = 16, CFO_PERT_WSUM_EIGEN_REDUX = 32, CFO_PERT_EIG_JACOBI = 64, CFO_PERT_NN_TIE_HIGH = 128,
       CFO_PERT_NN_TIE_FLANN = 256 };
typedef void (*cfo_voxel_sorter)(uint32_t* voxel_idx, uint32_t* point_idx, int n);
void cfo_set_perturbation(unsigned mask, uint64_t seed);
/* CFO_PERT_NN_TIE_FLANN's search on its own: the 1-NN of nq queries (x, y floats) among n 2-D float points, by the restated
 * flann::KDTreeSingleIndex (leaf 15, reorder, eps 0, KNNSimpleResultSet of one); idx_out / dist_out: nq entries */
void cfo_flann_nearest(const float* pts, int n, const float* queries, int nq, int* idx_out, float* dist_out);
unsigned cfo_get_perturbation(void);
void cfo_set_voxel_sorter(cfo_voxel_sorter fn);

/* Packed k-strongest slot: bits 0..15 range bin, 16..23 intensity, 24 valid, 25 peak. */
#define CFO_SLOT_RANGE(s) ((int)((s) & 0xFFFFu))
#define CFO_SLOT_INTENSITY(s) ((int)(((s) >> 16) & 0xFFu))
#define CFO_SLOT_VALID(s) ((int)(((s) >> 24) & 1u))
#define CFO_SLOT_PEAK(s) ((int)(((s) >> 25) & 1u))

/* Stage 1: StructuredKStrongest::FilterKstrongest (radar_filters.cpp:209-237) +
 * AxialNonMaxSupress (:238-298). img = A rows (azimuth) x R cols (range), row stride R.
 * out = A*k packed slots, per row ascending (intensity, range), unused slots 0. */
int cfo_filter(const uint8_t* img, int A, int R, int z_min, int k, uint32_t* out);

/* Independent cross-check of the top-k rule (full sort, no incremental insert). */
int cfo_filter_bruteforce(const uint8_t* img, int A, int R, int z_min, int k, uint32_t* out);

/* getPeaksFilteredPointCloud (radar_filters.cpp:309-337). peaks!=0 -> only slots with the peak flag.
 * xyi = 3 floats per point (x, y, intensity); returns number of points. */
/* azimuth CA-CFAR, the alternative stage-1 filter (cfar.cpp:27-87, radar_driver.cpp:52-56); returns the number
 * of detections, writes at most cap of them */
/* rho, rho', rho'' of the loss registration.cpp:78-97 builds (Ceres 2.0 forms; [3P]) */
void cfo_loss_eval(int loss, double loss_limit, double s, double rho[3]);
double cfo_cfar_scaling(int window_size, double false_alarm_rate);
int cfo_cfar(const uint8_t* img, int A, int R, float range_res, float static_threshold, float min_distance,
             double max_distance, int window_size, int nb_guard_cells, float false_alarm_rate, float* xyi, int cap);
int cfo_cfar_prefix(const uint8_t* img, int A, int R, float range_res, float static_threshold, float min_distance,
             double max_distance, int window_size, int nb_guard_cells, float false_alarm_rate, float* xyi, int cap);  /* the same decisions, window sums off a prefix sum */
int cfo_cloud(const uint32_t* slots, int A, int k, float range_res, float min_distance, int peaks,
              float* xyi);

/* Compensate (utils.cpp:96-113), mot = (tx, ty, theta) of the previous inter-frame motion. */
void cfo_compensate(float* xyi, int n, const double mot[3], int ccw);

/* One oriented surface point ("cell", pointnormal.h:45-105). */
typedef struct cfo_cell {
  double mean[2];
  double cov[3]; /* xx, xy, yy */
  double normal[2];
  double orth[2];
  double lambda_min, lambda_max;
  double scale; /* planarity, pointnormal.cpp:57 */
  double sum_intensity, avg_intensity;
  int32_t nsamples;
  int32_t valid;
} cfo_cell;

typedef struct cfo_scan cfo_scan; /* MapPointNormal */

/* MapPointNormal::MapPointNormal(cloud, radius, origin=(0,0), weight_intensity, raw=false)
 * (pointnormal.cpp:65-90, :265-297). brute!=0 disables the grid acceleration (same results). */
cfo_scan* cfo_scan_create(const float* xyi, int n, const cfo_params* p, int brute);
void cfo_scan_free(cfo_scan* s);
int cfo_scan_size(const cfo_scan* s);
const cfo_cell* cfo_scan_cells(const cfo_scan* s);
int cfo_scan_num_samples(const cfo_scan* s);           /* voxel centroids before the >=6 / valid cut */
const float* cfo_scan_samples(const cfo_scan* s);      /* 3 floats (x,y,intensity) per voxel centroid */
/* GetClosestIdx (pointnormal.cpp:238-254): -1 if none within d. */
int cfo_scan_closest(const cfo_scan* s, double px, double py, double d, int brute);

#define CFO_MAX_OUTER 64
typedef struct cfo_reg_summary {
  int32_t success;        /* Register() return value (covariance success) */
  int32_t usable;         /* loop 'success' flag before GetCovariance */
  int32_t outer_iterations; /* value of itr_ documented at n_scan_normal.cpp:161 */
  int32_t num_residuals;    /* of the last built problem */
  int32_t num_residual_blocks;
  int32_t reserved;
  double final_cost;      /* summary_.final_cost of the last solve */
  double score;           /* n_scan_normal.cpp:166 */
  int32_t inner_iterations[CFO_MAX_OUTER]; /* summary_.iterations.size() per outer iteration */
  int32_t termination[CFO_MAX_OUTER];      /* 0 convergence, 1 no_convergence, 2 failure */
  double outer_cost[CFO_MAX_OUTER];
  double outer_pose[CFO_MAX_OUTER][3];
} cfo_reg_summary;

/* n_scan_normal_reg::Register (n_scan_normal.cpp:82-187). poses = n x (x,y,theta) in/out;
 * cov6 = 36 doubles (row-major 6x6) of reg_cov.back(). */
/* Register with soft_constraints = true (n_scan_normal.cpp:373-377): a Mahalanobis prior on the last pose around its
 * initial value with information from prior_cov6 (reg_cov.back() as passed in), weighted by sqrt(#source cells) */
int cfo_register_soft(cfo_scan* const* scans, int n, double* poses_xyt, const double* prior_cov6, double* cov6,
                      const cfo_params* p, int brute, cfo_reg_summary* out);
/* n_scan_normal_reg::GetCost (n_scan_normal.cpp:188-213); itr = the object's itr_ (association radius, :222).
 * Returns the number of residuals or -1 (reference: false). */
int cfo_get_cost(cfo_scan* const* scans, int n, const double* poses_xyt, const cfo_params* p, int itr, int brute,
                 double* score, double* residuals, int cap);
/* OdometryKeyframeFuser::approximateCovarianceBySampling (odometrykeyframefuser.cpp:261-380); returns the
 * reference's bool, cov6 = 36 doubles row-major; sample_costs (optional) = steps^3 sampled costs */
int cfo_cov_by_sampling(cfo_scan* const* scans, int n, const double* poses_xyt, const cfo_params* p, int itr, int brute,
                        double xy_range, double yaw_range, int steps, double cov_scaler, double final_cost, int num_residuals,
                        double* cov6, double* sample_costs);
int cfo_register(cfo_scan* const* scans, int n, double* poses_xyt, double* cov6,
                 const cfo_params* p, int brute, cfo_reg_summary* out);

/* OdometryKeyframeFuser (odometrykeyframefuser.cpp:143-259) restated without ROS. */
typedef struct cfo_fuser cfo_fuser;
cfo_fuser* cfo_fuser_create(const cfo_params* p);
void cfo_fuser_free(cfo_fuser* f);
/* radarDriver::CallbackOffline + pointcloudCallback for one A x R polar sweep; pose_xyt out. */
int cfo_fuser_process_polar(cfo_fuser* f, const uint8_t* img, int A, int R, double pose_xyt[3]);
/* pointcloudCallback on an already filtered (uncompensated) cloud. */
int cfo_fuser_process_cloud(cfo_fuser* f, float* xyi, int n, double pose_xyt[3]);
void cfo_fuser_set_cov_sampling(cfo_fuser* f, int enable, double xy_range, double yaw_range, int steps, double scaler);
void cfo_fuser_last_cov(const cfo_fuser* f, double cov6[36]);
int cfo_fuser_num_keyframes(const cfo_fuser* f);
const cfo_reg_summary* cfo_fuser_last_summary(const cfo_fuser* f);
const cfo_scan* cfo_fuser_last_scan(const cfo_fuser* f);
/* stage timers in seconds, accumulated: [0] Filtering [1] compensate [2] build_normals [3] register */
void cfo_fuser_timers(const cfo_fuser* f, double t[4]);

#ifdef __cplusplus
}
#endif
#endif
