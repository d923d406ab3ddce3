/*
 * cfear_hip.h -- C ABI of the MI355X-native CFEAR hot path (libcfear_hip.so).
 *
 * Drop-in boundary for the per-scan path of dan11003/CFEAR_Radarodometry_code_public:
 *   k-strongest filter -> oriented surface points -> scan-to-keyframes registration.
 * Every entry point states the reference interface it replaces (file:line relative to the
 * reference repository). POD only: plain pointers, sizes and opaque handles; no C++/torch types.
 * All functions return 0 (CFEAR_OK) or a negative error code and never exit() (the reference
 * does: pointnormal.cpp:72-75, registration.cpp:24-25). cfear_last_error() gives the text.
 *
 * Threading: one cfear_ctx per caller thread / per sequence (the reference classes are not
 * re-entrant either, registration.h:104-131). All work of a context is ordered on its HIP stream.
 */
#ifndef CFEAR_HIP_H
#define CFEAR_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define CFEAR_OK 0
#define CFEAR_ERR_INVALID (-1)     /* bad argument */
#define CFEAR_ERR_HIP (-2)         /* HIP runtime failure */
#define CFEAR_ERR_UNSUPPORTED (-3) /* size outside the compiled kernel range */
#define CFEAR_ERR_EMPTY (-4)       /* empty cloud (reference: exit(0)) */
#define CFEAR_ERR_NOMEM (-5)
#define CFEAR_ERR_CAPACITY (-6)    /* a scan produced more oriented surface points than CFEAR_TUNE_MAX_CELLS allows */

/* registration.h:48-60 */
enum { CFEAR_COST_P2P = 0, CFEAR_COST_P2L = 1, CFEAR_COST_P2D = 2 };
enum { CFEAR_LOSS_NONE = 0, CFEAR_LOSS_HUBER = 1, CFEAR_LOSS_CAUCHY = 2, CFEAR_LOSS_SOFTLONE = 3,
       CFEAR_LOSS_COMBINED = 4, CFEAR_LOSS_TUKEY = 5 };

/* radar_driver.h:24 */
enum { CFEAR_FILTER_KSTRONG = 0, CFEAR_FILTER_CACFAR = 1 };

/* Union of radarDriver::Parameters (radar_driver.h:35-48), OdometryKeyframeFuser::Parameters
 * (odometrykeyframefuser.h:72-114) and the n_scan_normal_reg knobs (n_scan_normal.h:72-81,
 * n_scan_normal.cpp:9, registration.h:117-122) that are on the path. */
typedef struct cfear_params {
  float z_min;        /* radar_driver.h:40 (float, truncated to int at radar_filters.cpp:198) */
  float range_res;    /* radar_driver.h:41 */
  float min_distance; /* radar_driver.h:45 */
  int32_t k_strongest;      /* radar_driver.h:42; 1..64 supported */
  double res;               /* odometrykeyframefuser.h:97 (narrowed to float at pointnormal.h:118) */
  double downsample_factor; /* pointnormal.h:241 */
  int32_t weight_intensity; /* odometrykeyframefuser.h:92 */
  int32_t cost;             /* CFEAR_COST_*  (cost_type string, odometrykeyframefuser.h:86) */
  int32_t loss;             /* CFEAR_LOSS_*  (loss_type_ string, :99) */
  int32_t weight_opt;       /* registration.h:50 */
  double loss_limit;        /* :100 */
  double covar_scale;       /* :101, SetD2dPar n_scan_normal.h:53 */
  double regularization;    /* :102 */
  int32_t submap_scan_size; /* :91 */
  int32_t compensate;       /* :95 */
  int32_t radar_ccw;        /* :95 */
  int32_t use_keyframe;     /* :96 */
  double min_keyframe_dist;    /* :98 */
  double min_keyframe_rot_deg; /* :98 */
  int32_t max_itr_association; /* n_scan_normal.h:75 (8) */
  int32_t min_itr;             /* n_scan_normal.h:75 (3) */
  int32_t max_solver_iterations; /* n_scan_normal.cpp:9 (20) */
  int32_t filter_type;         /* CFEAR_FILTER_* (radarDriver::Parameters::filter_type_, radar_driver.h:24,48): the stage-1 filter of the
                                  batched odometry objects (cfear_odometry_*); the per-call entry points name their filter themselves */
  double assoc_radius;         /* registration.h:122 (2.0) */
  /* filter_type CA-CFAR (radar_driver.cpp:52-56: AzimuthCACFAR(window_size, false_alarm_rate, nb_guard_cells, range_res, z_min,
   * min_distance, 400.0)); z_min is the static threshold there, k_strongest is not used */
  int32_t cfar_window_size;    /* radar_driver.h:43 (10) */
  int32_t cfar_nb_guard_cells; /* radar_driver.h:43 (20) */
  float cfar_false_alarm_rate; /* radar_driver.h:44 (0.01) */
  int32_t cfar_max_points;     /* detections per sweep a batched odometry object is sized for (0 = 32768; a pcl cloud grows, device
                                  memory is sized up front: a sweep with more detections keeps the first ones in (azimuth, range) order
                                  and the reading calls fail with CFEAR_ERR_CAPACITY) */
  double cfar_max_distance;    /* radar_driver.cpp:54 (400.0) */
} cfear_params;

void cfear_default_params(cfear_params* p);

typedef struct cfear_ctx cfear_ctx;

/* One context = one HIP device + stream + scratch. `stream` may be NULL (the context creates its
 * own: note that the handle of a framework's *default* stream is NULL too, and the stream created here is not
 * ordered with that default stream) or an existing hipStream_t passed as void*: device buffers handed to the
 * *_device entry points must be complete on that stream (or synchronized) before the call. A, R = polar image shape the context is sized
 * for (rows = azimuths, cols = range bins; radar_driver.cpp:92-98). */
int cfear_create(cfear_ctx** out, int device, void* stream, const cfear_params* p, int A, int R);
void cfear_destroy(cfear_ctx* ctx);
const char* cfear_last_error(const cfear_ctx* ctx);
int cfear_set_params(cfear_ctx* ctx, const cfear_params* p);
int cfear_synchronize(cfear_ctx* ctx);

/* Launch-shape knobs of a context (tuning; an integration never needs them, tools/ and bench.py do). Results do not depend on
 * them. FILTER_OCCUPANCY: 5..7 filter waves per SIMD (default 7); FILTER_ROWS_PER_WAVE: consecutive azimuths walked by one
 * filter wave (0 = the default: 4, or 6 in launches of 1536 scans and more); ODOMETRY_OVERLAP = n (0..8): batched odometry objects created afterwards run the filter of a sweep
 * on a low-priority stream of their own, one sweep ahead, and the features / registration kernels of n contiguous ranges of
 * the sequences on n high-priority streams (0 = the three kernels strictly in turn on the context stream; DESIGN.md has the
 * measurements). REPLAY_PERSISTENT_MAX (default 256): cfear_odometry_replay_host runs up to this many sequences as persistent
 * workgroups that walk a whole chunk of sweeps in one launch; more sequences (or 0) take the two launches per sweep of the
 * batched step. FILTER_CUS = F (with ODOMETRY_OVERLAP >= 1): the filter stream is created with a compute-unit mask of F units spread
 * evenly over the chip and the odometry streams with the complement (hipExtStreamCreateWithCUMask), so that the HBM-bound filter
 * of sweep t + 1 and the latency-bound odometry kernels of sweep t run side by side without sharing a unit's registers and LDS.
 * REPEAT_SHORTCUT (default 1): an outer registration iteration (n_scan_normal.cpp:102-151) that starts from the pose, radius and
 * keyframes of the previous one - a solve that accepted no step - repeats it bit for bit, so its summary is taken from the
 * previous one instead of associating and solving again; 0 runs it again (the tests compare the two).
 * MAX_CELLS = n (default 0 = A * k_strongest, the number of filtered points: cannot overflow - or 4096 if that is less and
 * submap_scan_size > 7): oriented surface points per scan the batched odometry objects created afterwards are sized for. A sequence's
 * memory is (submap_scan_size + 1) scan blocks of ~12 B per point + 160 B per cell, plus submap_scan_size * cells * 76 B of
 * residual-block scratch: one cell per filtered point is 7 MB at submap_scan_size 4, k 12, but 200 MB at submap_scan_size 50, k 40
 * (launch/oxford_demo:62-71) - where real scans have a few hundred to ~1500 cells; hence the 4096 for the large submaps. A scan that produces more cells than n keeps the first n (ascending voxel index) and the reading
 * calls (poses / covariances / summary / replay_host) return CFEAR_ERR_CAPACITY from then on: never silently.
 * REGISTRATION_ORDER (default 1; 0 = in sequence order): batched odometry objects created afterwards hand their sequences to the registration workgroups
 * longest first - sorted on the device by the work each sequence's registration took in the previous sweep - so that the last
 * round of workgroups of a launch is not left to the slowest ones (results do not depend on it).
 * LARGE_SUBMAP_KERNEL (default 0): with submap_scan_size > 7 the batched step has two registration kernels - the production shape compiled for
 * 64 scans (256 threads, three workgroups per compute unit) and one that gives a registration a whole unit (512 threads, all of its LDS for
 * the residual blocks, every wave evaluating: 2 x faster per registration at fifty keyframes). 0 = the second when the sequences fit the
 * chip at one per unit or the submap has 24 keyframes or more, 1 = always the first, 2 = always the second. Poses agree to the summation order of the evaluation's partial sums.
 * NN_TIE_RULE (default 0) - the one knob here that DOES change results, on purpose: which of several exactly equidistant cells the 1-NN search
 * of GetClosestIdx (pointnormal.cpp:238-254) returns. The reference leaves that to FLANN's kd-tree; 1-5 % of a scan's cells share their float
 * mean with another cell, and the choice moves a registration by up to centimetres on a few percent of the sweeps of a cluttered scene
 * (DESIGN.md section 2). 0 = the lowest cell index (production: a uniform grid search); 1 = the highest (the other end, for sensitivity runs);
 * 2 = what a restatement of flann::KDTreeSingleIndex (leaf size 15, as pcl::KdTreeFLANN builds it) returns - scans and batched odometry objects
 * created afterwards also build that tree (one thread, ~0.1-0.2 ms per scan) and every association walks it pair by pair: a parity mode for
 * comparisons with a build of the reference, several times slower than production. Set it before the scans / objects are created.
 * VOXEL_ORDER (default 0) - the second such knob: the order in which the points of a VoxelGrid voxel are added up for its float centroid
 * (pointnormal.cpp:277-280). PCL sorts (voxel index, point) pairs on the voxel index alone with an UNSTABLE sort, so the order inside a voxel -
 * and the centroid's last bit, which now and then flips a point across the strict d^2 < r^2 of the radius search - belongs to the sort
 * routine: std::sort up to PCL 1.9 (Ubuntu 18.04), boost integer_sort from 1.10. 0 = by point index (production: what a stable sort gives);
 * 1 = exactly what libstdc++'s std::sort leaves, computed on the host by the same call on the same sequence - per-call scans only
 * (cfear_scan_create; batched odometry objects refuse the mode: it costs a host round trip per scan). */
enum { CFEAR_TUNE_FILTER_OCCUPANCY = 1, CFEAR_TUNE_FILTER_ROWS_PER_WAVE = 2, CFEAR_TUNE_ODOMETRY_OVERLAP = 3,
       CFEAR_TUNE_REPLAY_PERSISTENT_MAX = 4, CFEAR_TUNE_FILTER_CUS = 5, CFEAR_TUNE_REPEAT_SHORTCUT = 6, CFEAR_TUNE_MAX_CELLS = 7,
       CFEAR_TUNE_REGISTRATION_ORDER = 8, CFEAR_TUNE_LARGE_SUBMAP_KERNEL = 9, CFEAR_TUNE_NN_TIE_RULE = 10, CFEAR_TUNE_VOXEL_ORDER = 11 };
int cfear_tune(cfear_ctx* ctx, int key, int value);

/* ---- Stage 1: StructuredKStrongest (radar_filters.cpp:198-298) -----------------------------
 * Packed slot: bits 0..15 range bin | 16..23 intensity | 24 valid | 25 peak (AxialNonMaxSupress).
 * Per azimuth row k slots in ascending (intensity, range) order, unused slots = 0. */
#define CFEAR_SLOT_RANGE(s) ((int)((s) & 0xFFFFu))
#define CFEAR_SLOT_INTENSITY(s) ((int)(((s) >> 16) & 0xFFu))
#define CFEAR_SLOT_VALID(s) ((int)(((s) >> 24) & 1u))
#define CFEAR_SLOT_PEAK(s) ((int)(((s) >> 25) & 1u))

/* Batched filter on device-resident data, asynchronous on the context stream.
 * d_polar: n_scans contiguous A*R uint8 images; d_slots: n_scans*A*k uint32. */
int cfear_kstrongest_device(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, uint32_t* d_slots);
/* Same with host buffers (H2D, kernel, D2H, synchronous). Replaces FilterKstrongest +
 * AxialNonMaxSupress called from radarDriver::Process (radar_driver.cpp:58-60). */
int cfear_kstrongest_host(cfear_ctx* ctx, const uint8_t* h_polar, int n_scans, uint32_t* h_slots);

/* radarDriver::Callback for the non-Oxford datasets (radar_driver.cpp:74-90): the sensor image has rows = range bins and
 * columns = azimuths; cv::rotate(ROTATE_90_COUNTERCLOCKWISE) (:84) turns it into the rows = azimuth layout of every other
 * call here: out[i][j] = in[j][in_cols - 1 - i], out is in_cols x in_rows. The device variant takes n_images images back to
 * back and is asynchronous on the context stream; in and out must not overlap. */
int cfear_rotate_polar(cfear_ctx* ctx, const uint8_t* h_in, int in_rows, int in_cols, uint8_t* h_out);
int cfear_rotate_polar_device(cfear_ctx* ctx, const uint8_t* d_in, int n_images, int in_rows, int in_cols, uint8_t* d_out);

/* ---- Stage 1/1.5: point clouds -------------------------------------------------------------- */
typedef struct cfear_cloud cfear_cloud; /* pcl::PointCloud<pcl::PointXYZI> on the device */

/* radarDriver::CallbackOffline (radar_driver.cpp:163-176): polar image -> filtered cloud and
 * peaks cloud. h_polar: A*R uint8 host image (rows = azimuth). */
int cfear_filter_polar(cfear_ctx* ctx, const uint8_t* h_polar, cfear_cloud** cloud, cfear_cloud** cloud_peaks);
/* Same from a device-resident image (no copy). */
int cfear_filter_polar_device(cfear_ctx* ctx, const uint8_t* d_polar, cfear_cloud** cloud, cfear_cloud** cloud_peaks);
/* radarDriver::Process with filter_type "CA-CFAR" (radar_driver.cpp:52-56): AzimuthCACFAR(window_size,
 * false_alarm_rate, nb_guard_cells, range_res, z_min, min_distance, max_distance = 400.0)
 * .getFilteredPointCloud (cfar.cpp:27-87) -> filtered cloud, row-major over (azimuth, range bin). range_res, z_min
 * (static threshold) and min_distance come from the context parameters; the defaults of
 * radarDriver::Parameters are window_size 10, nb_guard_cells 20, false_alarm_rate 0.01 (radar_driver.h:43-44).
 * Synchronous (the cloud is sized by the detection count). */
int cfear_filter_cfar(cfear_ctx* ctx, const uint8_t* h_polar, int window_size, int nb_guard_cells, float false_alarm_rate,
                      double max_distance, cfear_cloud** cloud);
int cfear_filter_cfar_device(cfear_ctx* ctx, const uint8_t* d_polar, int window_size, int nb_guard_cells,
                             float false_alarm_rate, double max_distance, cfear_cloud** cloud);
/* The same filter over a batch of device-resident sweeps (n_scans images back to back), asynchronous on the context stream:
 * image i's detections go to d_xyi + i * capacity * 3 (x, y, intensity floats, same order as the per-call version, at most
 * `capacity` of them) and their number - the true count, which may exceed `capacity` - to d_counts[i]. The clouds feed
 * cfear_cloud_upload-style consumers or a caller's own kernels; sizes are known on the device without a host round trip. */
int cfear_filter_cfar_batch_device(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, int window_size, int nb_guard_cells,
                                   float false_alarm_rate, double max_distance, float* d_xyi, int capacity, int* d_counts);
/* Upload an existing cloud: xyi = n x (x, y, intensity) floats. */
int cfear_cloud_upload(cfear_ctx* ctx, const float* xyi, int n, cfear_cloud** cloud);
int cfear_cloud_size(cfear_ctx* ctx, const cfear_cloud* cloud, int* n);
int cfear_cloud_download(cfear_ctx* ctx, const cfear_cloud* cloud, float* xyi, int capacity, int* n);
/* m <= 16 clouds to the host with ONE synchronisation (counts and points travel together through pinned staging): what
 * radarDriver::CallbackOffline hands its caller is two clouds (radar_driver.cpp:59-60, 163-176), and Compensate (utils.cpp:96-113)
 * runs on both. xyi[i] receives min(n[i], capacity[i]) points; n[i] is the true count. xyi / capacity may be null (counts only). */
int cfear_clouds_download(cfear_ctx* ctx, const cfear_cloud* const* clouds, int m, float* const* xyi, const int* capacity, int* n);
/* Releasing a cloud or a scan parks its device block in the context for the next handle of a similar size (no hipFree, no
 * synchronisation on the per-sweep path; cfear_destroy gives everything back). Handles must not outlive their context. */
void cfear_cloud_release(cfear_ctx* ctx, cfear_cloud* cloud);
/* Compensate(cloud, Tmotion, ccw) (utils.h:49, utils.cpp:96-113); motion = (tx, ty, theta). In place. */
int cfear_compensate(cfear_ctx* ctx, cfear_cloud* cloud, const double motion_xyt[3], int ccw);
/* ... of the two clouds of one sweep by the same motion in one launch (odometrykeyframefuser.cpp:148-149 compensates cloud and cloud_peaks) */
int cfear_compensate_pair(cfear_ctx* ctx, cfear_cloud* cloud, cfear_cloud* cloud_peaks, const double motion_xyt[3], int ccw);

/* ---- Stage 2: MapPointNormal (pointnormal.h:110-243) ------------------------------------------ */
typedef struct cfear_scan cfear_scan; /* MapNormalPtr: cells + search structure, device resident */

/* One oriented surface point (class cell, pointnormal.h:45-105). */
typedef struct cfear_cell {
  double mean[2];   /* u_ */
  double cov[3];    /* cov_ xx, xy, yy */
  double normal[2]; /* snormal_ */
  double orth[2];   /* orth_normal (sign not defined by the reference) */
  double lambda_min, lambda_max;
  double scale;     /* scale_ (planarity) */
  double sum_intensity, avg_intensity;
  int32_t nsamples; /* Nsamples_ */
  int32_t valid;
} cfear_cell;

/* MapPointNormal(cld, radius = params.res, origin = (0,0), weight_intensity, raw = false)
 * (pointnormal.cpp:65-90 -> ComputeNormals :265-297 -> ComputeSearchTreeFromCells :151-162). */
int cfear_scan_create(cfear_ctx* ctx, const cfear_cloud* cloud, cfear_scan** scan);
/* MapPointNormal over given cells: raw = true (one identity cell per point, pointnormal.cpp:76-82, cell::GetIdentityCell
 * pointnormal.h:58,80-82) and the transformed-copy constructor (pointnormal.cpp:91-110) build the cell list on the host and
 * hand it over here; the device builds the search structure (ComputeSearchTreeFromCells :151-162) and the registration
 * views. Synchronous. */
int cfear_scan_from_cells(cfear_ctx* ctx, const cfear_cell* cells, int n, cfear_scan** scan);
void cfear_scan_release(cfear_ctx* ctx, cfear_scan* scan);
int cfear_scan_size(cfear_ctx* ctx, const cfear_scan* scan, int* n_cells);          /* GetSize() */
int cfear_scan_download_cells(cfear_ctx* ctx, const cfear_scan* scan, cfear_cell* cells, int capacity, int* n);
/* GetClosestIdx(p, d) (pointnormal.cpp:238-254) for nq query points; idx[i] = -1 if none. */
int cfear_scan_closest(cfear_ctx* ctx, const cfear_scan* scan, const double* qxy, int nq, double d, int32_t* idx);

/* ---- Stage 3: n_scan_normal_reg (n_scan_normal.h:27-85) ---------------------------------------- */
#define CFEAR_MAX_OUTER 64
typedef struct cfear_reg_summary {
  int32_t success;          /* Register() return value */
  int32_t usable;           /* solver solution usable (summary_.IsSolutionUsable()) */
  int32_t outer_iterations; /* itr_ as documented at n_scan_normal.cpp:161 */
  int32_t num_residuals;    /* problem_->NumResiduals() of the last problem */
  int32_t num_residual_blocks;
  int32_t assoc_path;       /* diagnostic, no reference counterpart: which association path the last problem build took - 1 = one block of source
                               cells against <= 4 keyframes (matches in registers), 2 = (group of four keyframes, cell) items dealt densely
                               (large submaps, dense scans), 3 = pair ranges per thread (the general path; also every NN_TIE_RULE != 0) */
  double final_cost;        /* summary_.final_cost */
  double score;             /* getScore() */
  int32_t inner_iterations[CFEAR_MAX_OUTER]; /* summary_.iterations.size() per outer iteration */
  int32_t termination[CFEAR_MAX_OUTER];      /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE */
  double outer_cost[CFEAR_MAX_OUTER];
  double outer_pose[CFEAR_MAX_OUTER][3];
} cfear_reg_summary;

/* bool Register(std::vector<MapNormalPtr>& scans, std::vector<Eigen::Affine3d>& Tsrc,
 *               std::vector<Matrix6d>& reg_cov, bool soft_constraints=false)
 * (n_scan_normal.cpp:82-187). scans[0..n-2] keyframes (fixed), scans[n-1] current.
 * poses_xyt: n x (x, y, theta) in/out; cov6_last: 36 doubles row-major = reg_cov.back(), in/out: a registration that
 * does not produce a usable solution leaves it as passed in, as the reference leaves reg_cov.
 * Returns CFEAR_OK; the reference's bool is summary->success. */
int cfear_register(cfear_ctx* ctx, cfear_scan* const* scans, int n, double* poses_xyt, double* cov6_last,
                   cfear_reg_summary* summary);

/* Register(..., soft_constraints = true) (n_scan_normal.cpp:373-377): adds the Mahalanobis prior
 * mahalanobisDistanceError (n_scan_normal.h:259-290) on the last pose around its initial value, with the information
 * matrix of prior_cov6 = reg_cov.back() as passed in (36 doubles row-major; rows/columns x, y, yaw are used,
 * registration.cpp:123-129) and the weight sqrt(#cells of the last scan). Everything else as cfear_register. */
int cfear_register_soft(cfear_ctx* ctx, cfear_scan* const* scans, int n, double* poses_xyt, const double* prior_cov6,
                        double* cov6_last, cfear_reg_summary* summary);

/* bool GetCost(std::vector<MapNormalPtr>& scans, std::vector<Eigen::Affine3d>& Tsrc, double& score,
 *              std::vector<double>& residuals) (n_scan_normal.cpp:188-213; called by the cost-sampling covariance,
 * odometrykeyframefuser.cpp:305): associations and residual blocks at the given poses, no solve. itr = the object's
 * itr_ (1 = double association radius, n_scan_normal.cpp:222). *score = 1/2 sum rho (ceres::Problem::Evaluate),
 * residuals = the robustified residuals in residual-block order (at most `capacity` are written), *n_residuals =
 * their number. Returns CFEAR_ERR_EMPTY (and *n_residuals = -1) where the reference returns false (<= 1 residuals). */
int cfear_get_cost(cfear_ctx* ctx, cfear_scan* const* scans, int n, const double* poses_xyt, int itr, double* score,
                   double* residuals, int capacity, int* n_residuals);

/* bool OdometryKeyframeFuser::approximateCovarianceBySampling(scans_vek, T_vek, cov_sampled)
 * (odometrykeyframefuser.cpp:261-380): GetCost on samples_per_axis^3 poses around poses_xyt[n-1] (all samples in one
 * launch, one workgroup each), least-squares quadratic, Hessian -> covariance scaled by GetCovarianceScaler
 * (final_cost / (num_residuals - 3) of the preceding Register) and covariance_scaler. Parameters are
 * par.cov_sampling_xy_range (0.4), cov_sampling_yaw_range (0.0043625), cov_sampling_samples_per_axis (3),
 * cov_sampling_covariance_scaler (4.0) (odometrykeyframefuser.h:107-110). *success = the reference's bool; cov6 = 36
 * doubles row-major (written only on success); sample_costs: optional samples_per_axis^3 sampled costs. */
int cfear_cov_by_sampling(cfear_ctx* ctx, cfear_scan* const* scans, int n, const double* poses_xyt, int itr, double xy_range,
                          double yaw_range, int samples_per_axis, double covariance_scaler, double final_cost,
                          int num_residuals, double* cov6, int* success, double* sample_costs);

/* ---- Batched odometry: OdometryKeyframeFuser::pointcloudCallback for B independent sequences ---
 * (odometrykeyframefuser.cpp:143-259, :397-411) with the filter of radar_driver.cpp:48-70 in front: k-strongest, or - with
 * cfear_params.filter_type = CFEAR_FILTER_CACFAR at creation - azimuth CA-CFAR (radar_driver.cpp:52-56; the peaks cloud stays empty
 * there and nothing on this path reads it).
 * All state (T_prev, Tmot, keyframe ring) lives on the device; one call = one radar sweep of every
 * sequence. */
typedef struct cfear_odometry cfear_odometry;
int cfear_odometry_create(cfear_ctx* ctx, int n_sequences, cfear_odometry** odo);
void cfear_odometry_destroy(cfear_ctx* ctx, cfear_odometry* odo);
int cfear_odometry_reset(cfear_ctx* ctx, cfear_odometry* odo);
/* d_polar: n_sequences contiguous A*R uint8 sweeps on the device. Asynchronous and stream-ordered for the input: the sweeps
 * must be ready at this point of the context stream, and work given to the context stream afterwards (the next write into
 * d_polar) runs after the filter has read it. With CFEAR_TUNE_ODOMETRY_OVERLAP the kernels run on two internal streams (the
 * filter of sweep t+1 beside the features / registration kernels of sweep t) that the context stream joins for the results only
 * in the reading calls below (poses / summary / profile_read / reset) and in cfear_synchronize(). */
int cfear_odometry_step_device(cfear_ctx* ctx, cfear_odometry* odo, const uint8_t* d_polar);
/* The same step from clouds that are already on the device instead of polar sweeps - whatever produced them (cfear_filter_cfar_batch_device,
 * a caller's own detector): sequence q's cloud is d_xyi + q * capacity * 3 (x, y, intensity floats, as cfear_cloud_upload takes them),
 * its size min(d_counts[q], capacity). What pointcloudCallback does with a cloud follows: Compensate (odometrykeyframefuser.cpp:146-150),
 * MapPointNormal (:161), Register, keyframe logic. capacity must not exceed the points the object's scans are sized for (A * k_strongest,
 * or cfar_max_points with filter_type CA-CFAR). Asynchronous on the context stream; d_xyi / d_counts are read by the first kernel only.
 * With filter_type CA-CFAR, cfear_odometry_step_device / _step_host / _replay_* run AzimuthCACFAR in front of exactly this. */
int cfear_odometry_step_cloud_device(cfear_ctx* ctx, cfear_odometry* odo, const float* d_xyi, int capacity, const int* d_counts);
/* Same from a host buffer (n_sequences * A * R bytes): the sweeps are copied to a staging buffer on the device; the call
 * returns when that copy has completed (h_polar may be reused or freed at once), the kernels run asynchronously as above. */
int cfear_odometry_step_host(cfear_ctx* ctx, cfear_odometry* odo, const uint8_t* h_polar);
/* Tcurrent of every sequence as (x, y, theta); synchronises the stream. */
int cfear_odometry_poses(cfear_ctx* ctx, cfear_odometry* odo, double* poses_xyt);
/* cov_current of every sequence after the last sweep (the Covariance& of the five-argument pointcloudCallback,
 * odometrykeyframefuser.cpp:196,397-411): n_sequences x 36 doubles row-major - the registration covariance of the sweep's
 * Register() (GetCovariance), the identity FormatScans starts from when the registration had no usable solution, zeros before
 * the second sweep. (The cost-sampling variant, estimate_cov_by_sampling, is the per-call cfear_cov_by_sampling.) Synchronises. */
int cfear_odometry_covariances(cfear_ctx* ctx, cfear_odometry* odo, double* cov6);
/* Has any scan of this object been truncated - more oriented surface points than CFEAR_TUNE_MAX_CELLS, or a cloud with more points than the object
 * holds? Synchronises the context stream; returns CFEAR_OK or CFEAR_ERR_CAPACITY (with the message the reading calls give). For callers of the
 * asynchronous cfear_odometry_replay_device, which read their records on the device and never pass through poses / summary / replay_host.
 * per_sequence (optional, n_sequences words): which sequences - bit 0 = a scan of that sequence lost cells, bit 1 = a cloud of it lost points;
 * the others' results are untouched (the condition is sticky until cfear_odometry_reset, like the return code). */
int cfear_odometry_status(cfear_ctx* ctx, cfear_odometry* odo, int32_t* per_sequence);
/* Last Register() summary / cell count / keyframe count of one sequence (debug + parity tests). */
int cfear_odometry_summary(cfear_ctx* ctx, cfear_odometry* odo, int sequence, cfear_reg_summary* summary,
                           int* n_cells, int* n_keyframes);

/* ---- Replay of a recording: the loop of offline_odometry.cpp:103-125 (read a sweep, radarDriver::CallbackOffline,
 * OdometryKeyframeFuser::pointcloudCallback, keep the pose) for n_sweeps consecutive sweeps without a host round trip per
 * sweep. h_frames: n_sweeps x n_sequences x A x R bytes (sweep-major: sweep t of every sequence, then sweep t + 1). The sweeps
 * are copied and filtered in chunks on a stream of their own, two chunks in flight (the filter needs nothing but its input,
 * radar_driver.cpp:58), while features -> registration run sweep after sweep on the context stream; what a caller of
 * pointcloudCallback could observe after every sweep is kept on the device and comes back once, at the end. The state of
 * `odo` carries over between calls (and to / from cfear_odometry_step_*): a long recording may be handed over in pieces.
 * Synchronous: returns when the last sweep is done. Copies from pinned memory (cfear_host_alloc) overlap with the kernels;
 * from pageable memory the call still works, each chunk's copy then holds the calling thread. */
typedef struct cfear_sweep_record {
  double pose[3];             /* Tcurrent after the sweep (x, y, theta) */
  double final_cost;          /* summary_.final_cost of the sweep's Register() (0 for the first sweep) */
  int32_t outer_iterations;   /* as cfear_reg_summary */
  int32_t num_residuals;
  int32_t n_keyframes;        /* keyframes after the sweep (odometrykeyframefuser.cpp:470-476) */
  int32_t n_cells;            /* oriented surface points of the sweep's scan */
  int32_t inner_iterations[8];/* summary_.iterations.size() of the first 8 outer iterations */
} cfear_sweep_record;
int cfear_odometry_replay_host(cfear_ctx* ctx, cfear_odometry* odo, const uint8_t* h_frames, int n_sweeps,
                               cfear_sweep_record* records /* n_sweeps x n_sequences, or NULL */);
/* The same for sweeps that are already on the device (n_sweeps x n_sequences x A x R bytes, ready at this point of the context
 * stream), ASYNCHRONOUS: the call returns when everything is queued; d_records (device memory, n_sweeps x n_sequences records, or
 * NULL) is filled by the kernels, the sweeps are read until the last chunk's filter has run - work queued on the context stream
 * afterwards (a copy of d_records, the next write into d_frames) is ordered behind all of it. */
int cfear_odometry_replay_device(cfear_ctx* ctx, cfear_odometry* odo, const uint8_t* d_frames, int n_sweeps,
                                 cfear_sweep_record* d_records);
/* Page-locked host memory for the sweeps of a replay (hipHostMalloc / hipHostFree). */
int cfear_host_alloc(cfear_ctx* ctx, size_t bytes, void** out);
void cfear_host_free(cfear_ctx* ctx, void* p);

/* Filter-kernel timing with HIP events (bench.py roofline leg): enable, run steps, read. filter_seconds is the sum
 * of the durations of the filter launches, each measured on the stream it ran on. The events come from a pool created
 * when profiling is enabled (and grown in blocks), not one hipEventCreate per launch. */
int cfear_odometry_profile(cfear_ctx* ctx, cfear_odometry* odo, int enable);
int cfear_odometry_profile_read(cfear_ctx* ctx, cfear_odometry* odo, double* filter_seconds, int* filter_launches);
/* The same for the two kernels behind the filter (stages 1.5-2: cloud + compensation + oriented surface points;
 * stage 3 + caller: registration and keyframe logic), summed over the profiled launches. */
int cfear_odometry_profile_read_stages(cfear_ctx* ctx, cfear_odometry* odo, double* features_seconds, double* registration_seconds,
                                       int* launches);

/* Per-workgroup phase timestamps (tuning / bench.py's workgroup-time percentiles): enable = 1 switches the odometry steps
 * to the timed instantiations of the features and registration kernels (thread 0 of every workgroup stores wall_clock64()
 * ticks of 10 ns: slots 0..13 features phases, 14..28 registration phases, 29..31 accumulated evaluation / controller
 * ticks and evaluation count - these three need clock reads around every LM command and halve the speed of the
 * registration kernel; enable = 2 leaves them out: phase stamps only, still the timed instantiations; enable = 3 keeps the
 * PRODUCTION kernels and only records every workgroup's start and end clock in slots 0, 1 and 14, 15 - what bench.py's
 * workgroup-time percentiles use; enable = 4: timed instantiations, and slots 0..7 hold - instead of the first feature
 * stamps - the breakdown of the registration's command loop over its evaluation commands: ticks waiting at the command
 * barrier, executing the command, waiting at the result barrier, the command count, ticks in the controller's state
 * function (which includes the trust-region step that follows it), in trust-region steps taken on their own, in the
 * publishing function and in the end-of-solve function); with host_ticks != NULL the [n_sequences][32] table of the steps since the last read is
 * copied out (synchronises) and cleared. enable = 0 frees the table: back to the production kernels. */
int cfear_odometry_phase_times(cfear_ctx* ctx, cfear_odometry* odo, int enable, long long* host_ticks);

/* Timing hook used by bench.py: seconds of the filter kernel measured with HIP events on the
 * context stream over `iters` launches (after `warmup`). */
int cfear_time_kstrongest(cfear_ctx* ctx, const uint8_t* d_polar, int n_scans, uint32_t* d_slots,
                          int warmup, int iters, double* avg_seconds);

const char* cfear_version(void);

#ifdef __cplusplus
}
#endif
#endif
