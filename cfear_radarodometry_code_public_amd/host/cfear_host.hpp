// cfear_host.hpp -- C++ host-side mirror of the reference's interfaces for the hot path, on top of
// the C ABI (include/cfear_hip.h). Same class names, argument meaning and call structure as
//   radarDriver            (radar_driver.h:32-120,   radar_driver.cpp:23-176)
//   Compensate             (utils.h:49,               utils.cpp:96-113)
//   MapPointNormal         (pointnormal.h:110-243,   pointnormal.cpp:65-90, 238-254)
//   n_scan_normal_reg      (n_scan_normal.h:27-85,   n_scan_normal.cpp:82-187)
//   OdometryKeyframeFuser  (odometrykeyframefuser.h:67-260, odometrykeyframefuser.cpp:143-259)
// with POD stand-ins where the reference uses ROS / PCL / Eigen / OpenCV types (none of those are in
// this image; INTEGRATION.md shows the two-line adapters for a tree that has them).
// Error behaviour: where the reference prints and calls exit(0) this layer throws std::runtime_error
// carrying cfear_last_error(); bool returns are kept.
#pragma once
#include <cmath>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/cfear_hip.h"

namespace CFEAR_Radarodometry {

// ---- POD stand-ins ---------------------------------------------------------------------------
struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; };                 // pcl::PointXYZI
struct PointCloudXYZI { std::vector<PointXYZI> points; uint64_t stamp = 0;       // pcl::PointCloud<pcl::PointXYZI>
  size_t size() const { return points.size(); } };
typedef std::shared_ptr<PointCloudXYZI> CloudPtr;                                // ...::Ptr
struct PolarImage { int rows = 0, cols = 0; const uint8_t* data = nullptr; uint64_t stamp = 0; };  // sensor_msgs::Image 8UC1 (rows = azimuth)
struct Vector2d { double x = 0, y = 0; double operator()(int i) const { return i ? y : x; } };
struct Matrix2d { double m[2][2] = {{0, 0}, {0, 0}}; double operator()(int r, int c) const { return m[r][c]; } };
typedef struct Matrix6dT { double m[6][6]; Matrix6dT() { for (auto& r : m) for (double& v : r) v = 0; for (int i = 0; i < 6; i++) m[i][i] = 1; } } Matrix6d;

// Eigen::Affine3d restricted to what the path uses: planar rigid motions (vectorToAffine3d, registration.cpp:130-136)
struct Affine3d {
  double l[2][2] = {{1, 0}, {0, 1}}, t[2] = {0, 0};
  static Affine3d Identity() { return Affine3d(); }
  static Affine3d FromXYT(double x, double y, double th) { Affine3d T; const double c = std::cos(th), s = std::sin(th); T.l[0][0] = c; T.l[0][1] = -s; T.l[1][0] = s; T.l[1][1] = c; T.t[0] = x; T.t[1] = y; return T; }
  Affine3d operator*(const Affine3d& B) const { Affine3d C; for (int i = 0; i < 2; i++) { for (int j = 0; j < 2; j++) C.l[i][j] = l[i][0] * B.l[0][j] + l[i][1] * B.l[1][j]; C.t[i] = (l[i][0] * B.t[0] + l[i][1] * B.t[1]) + t[i]; } return C; }
  Affine3d inverse() const { Affine3d I; const double det = l[0][0] * l[1][1] - l[0][1] * l[1][0], id = 1.0 / det; I.l[0][0] = l[1][1] * id; I.l[0][1] = -l[0][1] * id; I.l[1][0] = -l[1][0] * id; I.l[1][1] = l[0][0] * id; I.t[0] = -(I.l[0][0] * t[0] + I.l[0][1] * t[1]); I.t[1] = -(I.l[1][0] * t[0] + I.l[1][1] * t[1]); return I; }
  double translation_norm() const { return std::sqrt(t[0] * t[0] + t[1] * t[1]); }
  double yaw() const { return std::atan2(l[1][0], l[1][1]); }  // eulerAngles(0,1,2)[2] of a pure yaw rotation
};
inline void Affine3dToVectorXYeZ(const Affine3d& T, std::vector<double>& par) { par.resize(3); par[0] = T.t[0]; par[1] = T.t[1]; par[2] = T.yaw(); }  // utils.cpp:115-122
inline Affine3d vectorToAffine3d(const std::vector<double>& v) { return Affine3d::FromXYT(v[0], v[1], v[2]); }

typedef enum costmetric { P2P, P2L, P2D } cost_metric;                                        // registration.h:55
typedef enum losstype { None, Huber, Cauchy, SoftLOne, Combined, Tukey } loss_type;          // registration.h:60
typedef enum weight_options { Uniform = 0, Sim_N = 1, Sim_direciton = 2, Sim_scale = 3, Combined_weights = 4 } weightoption;  // :50
inline cost_metric Str2Cost(const std::string& s) { return s == "P2L" ? P2L : (s == "P2D" ? P2D : P2P); }  // registration.cpp:30-37
inline loss_type Str2loss(const std::string& s) {                                              // registration.cpp:50-65
  if (s == "Cauchy") return Cauchy; if (s == "SoftLOne") return SoftLOne; if (s == "Combined") return Combined;
  if (s == "Tukey") return Tukey; if (s == "None") return None; return Huber; }

// ---- timing (statistics.h / statistics.cpp:10-51): same stage names as the reference ------------
struct statistics {
  std::map<std::string, std::vector<double>> executionTimes;
  void Document(const std::string& name, double value) { executionTimes[name].push_back(value); }
  std::string GetStatistics() const { std::string s; for (auto& kv : executionTimes) { double m = 0; for (double v : kv.second) m += v; m /= kv.second.empty() ? 1 : kv.second.size(); s += kv.first + ", mean " + std::to_string(m) + ", count " + std::to_string(kv.second.size()) + "\n"; } return s; }
};
inline statistics& timing_instance() { static statistics t; return t; }
#define CFEAR_TIMING CFEAR_Radarodometry::timing_instance()

// ---- one device context shared by the mirrored classes (one per thread / sequence) -------------
class Device {
 public:
  Device(const cfear_params& p, int A, int R, int device = 0) : par_(p), A_(A), R_(R) {
    if (cfear_create(&ctx_, device, nullptr, &par_, A, R) != CFEAR_OK) throw std::runtime_error("cfear_create failed: no usable gfx950 device or invalid parameters");
  }
  ~Device() { cfear_destroy(ctx_); }
  Device(const Device&) = delete;
  cfear_ctx* ctx() const { return ctx_; }
  const cfear_params& params() const { return par_; }
  void set_params(const cfear_params& p) { check(cfear_set_params(ctx_, &p), "cfear_set_params"); par_ = p; }
  void check(int rc, const char* what) const { if (rc != CFEAR_OK) throw std::runtime_error(std::string(what) + ": " + cfear_last_error(ctx_)); }
  int A() const { return A_; } int R() const { return R_; }
 private:
  cfear_ctx* ctx_ = nullptr; cfear_params par_; int A_, R_;
};
typedef std::shared_ptr<Device> DevicePtr;

// device-resident cloud handle travelling with the host cloud (keeps the data on the GPU between stages)
struct DeviceCloud { DevicePtr dev; cfear_cloud* h = nullptr; ~DeviceCloud() { if (h) cfear_cloud_release(dev->ctx(), h); } };
typedef std::shared_ptr<DeviceCloud> DeviceCloudPtr;

inline CloudPtr DownloadCloud(const DevicePtr& dev, cfear_cloud* h) {
  int n = 0; dev->check(cfear_cloud_size(dev->ctx(), h, &n), "cfear_cloud_size");
  std::vector<float> xyi(3 * (size_t)(n > 0 ? n : 1));
  dev->check(cfear_cloud_download(dev->ctx(), h, xyi.data(), n, &n), "cfear_cloud_download");
  CloudPtr c(new PointCloudXYZI()); c->points.resize(n);
  for (int i = 0; i < n; i++) { c->points[i].x = xyi[3 * i]; c->points[i].y = xyi[3 * i + 1]; c->points[i].intensity = xyi[3 * i + 2]; }
  return c;
}
inline DeviceCloudPtr UploadCloud(const DevicePtr& dev, const PointCloudXYZI& c) {
  std::vector<float> xyi(3 * c.size() + 3);
  for (size_t i = 0; i < c.size(); i++) { xyi[3 * i] = c.points[i].x; xyi[3 * i + 1] = c.points[i].y; xyi[3 * i + 2] = c.points[i].intensity; }
  DeviceCloudPtr d(new DeviceCloud()); d->dev = dev;
  dev->check(cfear_cloud_upload(dev->ctx(), xyi.data(), (int)c.size(), &d->h), "cfear_cloud_upload");
  return d;
}

// ---- radarDriver (radar_driver.h:32-120) ----------------------------------------------------------
typedef enum filter_type { kstrong, CACFAR } filtertype;  // radar_driver.h:24
inline filtertype Str2filter(const std::string& str) { return str == "CA-CFAR" ? filtertype::CACFAR : filtertype::kstrong; }  // radar_driver.cpp:6-12
inline std::string Filter2str(const filtertype& filter) { return filter == filtertype::CACFAR ? "CA-CFAR" : "kstrong"; }       // :13-20

// ---- AzimuthCACFAR (cfar.h:31-46, cfar.cpp:27-87) -------------------------------------------------------
class AzimuthCACFAR {
 public:
  AzimuthCACFAR(const DevicePtr& dev, int window_size = 40, double false_alarm_rate = 0.01, int nb_guard_cells = 5, double range_resolution = 0.0438,
                double static_threshold = 60.0, double min_distance = 2.5, double max_distance = 200.0)
      : dev_(dev), window_size_(window_size), nb_guard_cells_(nb_guard_cells), false_alarm_rate_(false_alarm_rate), max_distance_(max_distance) {
    cfear_params p = dev_->params(); p.range_res = (float)range_resolution; p.z_min = (float)static_threshold; p.min_distance = (float)min_distance;
    dev_->set_params(p);
  }
  // void getFilteredPointCloud(const cv_bridge::CvImagePtr&, PointCloud::Ptr& output) const (cfar.cpp:35)
  DeviceCloudPtr getFilteredPointCloud(const PolarImage& img, CloudPtr& output_pointcloud) const {
    DeviceCloudPtr d(new DeviceCloud()); d->dev = dev_;
    dev_->check(cfear_filter_cfar(dev_->ctx(), img.data, window_size_, nb_guard_cells_, (float)false_alarm_rate_, max_distance_, &d->h), "cfear_filter_cfar");
    output_pointcloud = DownloadCloud(dev_, d->h);
    return d;
  }
 private:
  DevicePtr dev_; int window_size_, nb_guard_cells_; double false_alarm_rate_, max_distance_;
};

class radarDriver {
 public:
  class Parameters {  // radar_driver.h:35-84
   public:
    float z_min = 60; float range_res = 0.0438f; int azimuths = 400, k_strongest = 12;
    float min_distance = 2.5f, max_distance = 200; std::string dataset = "oxford";
    int nb_guard_cells = 20, window_size = 10; float false_alarm_rate = 0.01f;  // :43-44
    filtertype filter_type_ = filtertype::kstrong;                               // :48
  };
  radarDriver(const DevicePtr& dev, const Parameters& pars, bool disable_callback = true) : dev_(dev), par(pars) {
    (void)disable_callback;
    cfear_params p = dev_->params(); p.z_min = par.z_min; p.range_res = par.range_res; p.min_distance = par.min_distance; p.k_strongest = par.k_strongest;
    dev_->set_params(p);
  }
  // void CallbackOffline(const sensor_msgs::ImageConstPtr&, PointCloud::Ptr& cloud, PointCloud::Ptr& cloud_peaks) (radar_driver.cpp:163-176)
  // -> Process() (:48-73). The image is rows = azimuth x cols = range (the reference rotates non-Oxford input first, :84).
  void CallbackOffline(const PolarImage& img, CloudPtr& cloud, CloudPtr& cloud_peaks) {
    if (!img.data) throw std::runtime_error("Radar image NULL");  // radar_driver.cpp:75-78
    if (img.rows != dev_->A() || img.cols != dev_->R()) throw std::runtime_error("polar image shape differs from the device context");
    cv_polar_image = img;
    if (par.filter_type_ == filtertype::CACFAR) {  // :52-56: max_distance 400.0, the peaks cloud stays empty
      AzimuthCACFAR filter(dev_, par.window_size, par.false_alarm_rate, par.nb_guard_cells, par.range_res, par.z_min, par.min_distance, 400.0);
      last_cloud_ = filter.getFilteredPointCloud(img, cloud);
      last_peaks_.reset(new DeviceCloud()); last_peaks_->dev = dev_;
      cloud_peaks.reset(new PointCloudXYZI());
    } else {
      last_cloud_.reset(new DeviceCloud()); last_peaks_.reset(new DeviceCloud());
      last_cloud_->dev = dev_; last_peaks_->dev = dev_;
      dev_->check(cfear_filter_polar(dev_->ctx(), img.data, &last_cloud_->h, &last_peaks_->h), "cfear_filter_polar");
      cloud = DownloadCloud(dev_, last_cloud_->h); cloud_peaks = DownloadCloud(dev_, last_peaks_->h);
    }
    cloud->stamp = cloud_peaks->stamp = img.stamp;
  }
  // void Callback(const sensor_msgs::ImageConstPtr&) for the non-Oxford datasets (radar_driver.cpp:74-90): the image arrives with
  // rows = range bins; cv::rotate(ROTATE_90_COUNTERCLOCKWISE) (:84) runs on the device, then Process()
  void Callback(const PolarImage& range_major, CloudPtr& cloud, CloudPtr& cloud_peaks) {
    if (!range_major.data) throw std::runtime_error("Radar image NULL");
    rotated_.resize((size_t)range_major.rows * range_major.cols);
    dev_->check(cfear_rotate_polar(dev_->ctx(), range_major.data, range_major.rows, range_major.cols, rotated_.data()), "cfear_rotate_polar");
    PolarImage az; az.rows = range_major.cols; az.cols = range_major.rows; az.data = rotated_.data(); az.stamp = range_major.stamp;
    CallbackOffline(az, cloud, cloud_peaks);
  }
  DeviceCloudPtr device_cloud() const { return last_cloud_; }  // avoids a round trip when the fuser runs on the same device
  PolarImage cv_polar_image;  // latest radar image (radar_driver.h:92)
 private:
  DevicePtr dev_; Parameters par; DeviceCloudPtr last_cloud_, last_peaks_; std::vector<uint8_t> rotated_;
};

// ---- Compensate (utils.h:49) -------------------------------------------------------------------------
inline void Compensate(const DevicePtr& dev, DeviceCloud& cloud, const Affine3d& Tmotion, bool ccw) {
  const double mot[3] = {Tmotion.t[0], Tmotion.t[1], Tmotion.yaw()};
  dev->check(cfear_compensate(dev->ctx(), cloud.h, mot, ccw ? 1 : 0), "cfear_compensate");
}

// ---- cell / MapPointNormal (pointnormal.h:45-243) -----------------------------------------------------
struct cell {
  Vector2d u_; Matrix2d cov_; double scale_ = 0; Vector2d snormal_, orth_normal; double lambda_min = 0, lambda_max = 0;
  double sum_intensity_ = 0, avg_intensity_ = 0; size_t Nsamples_ = 0; bool valid_ = false;
  double GetPlanarity() const { return scale_; }
};
class MapPointNormal;
typedef std::shared_ptr<MapPointNormal> MapNormalPtr;
class MapPointNormal {
 public:
  // MapPointNormal(cld, radius, origin = (0,0), weight_intensity = false, raw = false) (pointnormal.h:118)
  MapPointNormal(const DevicePtr& dev, const DeviceCloud& cld, float radius, const Vector2d& origin = Vector2d(), bool weight_intensity = false, bool raw = false) : dev_(dev) {
    if (raw) throw std::runtime_error("MapPointNormal: raw=true (identity cells) is not on the accelerated path");
    if (origin.x != 0 || origin.y != 0) throw std::runtime_error("MapPointNormal: origin must be (0,0) as in odometrykeyframefuser.cpp:161");
    cfear_params p = dev_->params(); p.res = radius; p.weight_intensity = weight_intensity ? 1 : 0; dev_->set_params(p);
    const int rc = cfear_scan_create(dev_->ctx(), cld.h, &scan_);
    if (rc == CFEAR_ERR_EMPTY) throw std::runtime_error("error, cloud empty");  // pointnormal.cpp:72-75 (exit(0) there)
    dev_->check(rc, "cfear_scan_create");
  }
  ~MapPointNormal() { if (scan_) cfear_scan_release(dev_->ctx(), scan_); }
  size_t GetSize() { int n = 0; dev_->check(cfear_scan_size(dev_->ctx(), scan_, &n), "cfear_scan_size"); return (size_t)n; }
  const std::vector<cell>& GetCells() { fetch(); return cells_; }
  cell& GetCell(size_t i) { fetch(); return cells_[i]; }
  Vector2d GetMean2d(size_t i) { fetch(); return cells_[i].u_; }
  Matrix2d GetCov2d(size_t i) { fetch(); return cells_[i].cov_; }
  Vector2d GetNormal2d(size_t i) { fetch(); return cells_[i].snormal_; }
  std::vector<int> GetClosestIdx(const Vector2d& p, double d) {  // pointnormal.cpp:238-254
    const double q[2] = {p.x, p.y}; int32_t idx = -1;
    dev_->check(cfear_scan_closest(dev_->ctx(), scan_, q, 1, d, &idx), "cfear_scan_closest");
    return idx >= 0 ? std::vector<int>{idx} : std::vector<int>();
  }
  // double GetCellRelTimeStamp(index, ccw) (pointnormal.cpp:139-143) with GetRelTimeStamp (utils.h:28-32)
  double GetCellRelTimeStamp(size_t index, bool ccw) {
    fetch();
    const double a = std::atan2(cells_[index].u_.y, cells_[index].u_.x);
    const double d = (a > 0.00001 ? a : (2 * M_PI + a)) / (2 * M_PI);
    return ccw ? -(d - 0.5) : (d - 0.5);
  }
  // std::vector<cell> TransformCells(T) (pointnormal.cpp:352-361) -> cell::TransformCopy (:515-527), arithmetic as written
  // there: C = R * T * cov * R^T with the affine T applied to the columns of cov (so the translation enters the product)
  std::vector<cell> TransformCells(const Affine3d& T) {
    fetch();
    std::vector<cell> out;
    const double (*R)[2] = T.l;
    for (const cell& c : cells_) {
      cell ct = c;
      double M[2][2], RT[2][2], rt[2];  // R*T: linear R*R, translation R*t
      for (int i = 0; i < 2; i++) { for (int j = 0; j < 2; j++) RT[i][j] = R[i][0] * R[0][j] + R[i][1] * R[1][j]; rt[i] = R[i][0] * T.t[0] + R[i][1] * T.t[1]; }
      for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) M[i][j] = RT[i][0] * c.cov_.m[0][j] + RT[i][1] * c.cov_.m[1][j] + rt[i];
      for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) ct.cov_.m[i][j] = M[i][0] * R[j][0] + M[i][1] * R[j][1];
      ct.snormal_.x = R[0][0] * c.snormal_.x + R[0][1] * c.snormal_.y; ct.snormal_.y = R[1][0] * c.snormal_.x + R[1][1] * c.snormal_.y;
      ct.orth_normal.x = R[0][0] * c.orth_normal.x + R[0][1] * c.orth_normal.y; ct.orth_normal.y = R[1][0] * c.orth_normal.x + R[1][1] * c.orth_normal.y;
      ct.u_.x = R[0][0] * c.u_.x + R[0][1] * c.u_.y + T.t[0]; ct.u_.y = R[1][0] * c.u_.x + R[1][1] * c.u_.y + T.t[1];
      out.push_back(ct);
    }
    return out;
  }
  cfear_scan* handle() const { return scan_; }
  static double downsample_factor;  // pointnormal.h:241 (set through cfear_params.downsample_factor)
 private:
  void fetch() {
    if (fetched_) return;
    int n = 0; dev_->check(cfear_scan_size(dev_->ctx(), scan_, &n), "cfear_scan_size");
    std::vector<cfear_cell> raw((size_t)(n > 0 ? n : 1));
    dev_->check(cfear_scan_download_cells(dev_->ctx(), scan_, raw.data(), n, &n), "cfear_scan_download_cells");
    cells_.resize(n);
    for (int i = 0; i < n; i++) {
      cell& c = cells_[i]; const cfear_cell& r = raw[i];
      c.u_.x = r.mean[0]; c.u_.y = r.mean[1]; c.cov_.m[0][0] = r.cov[0]; c.cov_.m[0][1] = c.cov_.m[1][0] = r.cov[1]; c.cov_.m[1][1] = r.cov[2];
      c.snormal_.x = r.normal[0]; c.snormal_.y = r.normal[1]; c.orth_normal.x = r.orth[0]; c.orth_normal.y = r.orth[1];
      c.lambda_min = r.lambda_min; c.lambda_max = r.lambda_max; c.scale_ = r.scale; c.sum_intensity_ = r.sum_intensity; c.avg_intensity_ = r.avg_intensity;
      c.Nsamples_ = (size_t)r.nsamples; c.valid_ = r.valid != 0;
    }
    fetched_ = true;
  }
  DevicePtr dev_; cfear_scan* scan_ = nullptr; std::vector<cell> cells_; bool fetched_ = false;
};
inline double MapPointNormal::downsample_factor = 1;

// ---- n_scan_normal_reg (n_scan_normal.h:27-85) ----------------------------------------------------------
class n_scan_normal_reg {
 public:
  n_scan_normal_reg(const DevicePtr& dev, const cost_metric& cost, loss_type loss = Huber, double loss_limit = 0.1, const weightoption opt = Uniform)
      : dev_(dev), cost_(cost), loss_(loss), loss_limit_(loss_limit), weight_opt_(opt) {}
  void SetD2dPar(const double cov_scale, const double regularization) { cov_scale_ = cov_scale; regularization_ = regularization; }  // n_scan_normal.h:53
  void SetParameters(unsigned int max_itr_association, unsigned int max_itr_solver) { max_itr_association_ = (int)max_itr_association; max_itr_solver_ = (int)max_itr_solver; }
  // bool Register(scans, Tsrc, reg_cov, soft_constraints = false) (n_scan_normal.cpp:82-187); only Tsrc.back() is free.
  bool Register(std::vector<MapNormalPtr>& scans, std::vector<Affine3d>& Tsrc, std::vector<Matrix6d>& reg_cov, bool soft_constraints = false) {
    const size_t n = scans.size();
    if (Tsrc.size() != n || reg_cov.size() != n || n < 2) throw std::runtime_error("Register: scans/Tsrc/reg_cov size mismatch");  // assert at n_scan_normal.cpp:84
    cfear_params p = dev_->params();
    p.cost = cost_ == P2L ? CFEAR_COST_P2L : (cost_ == P2D ? CFEAR_COST_P2D : CFEAR_COST_P2P); p.loss = (int)loss_; p.loss_limit = loss_limit_;
    p.weight_opt = (int)weight_opt_; p.covar_scale = cov_scale_; p.regularization = regularization_;
    p.max_itr_association = max_itr_association_; p.max_solver_iterations = max_itr_solver_;
    dev_->set_params(p);
    std::vector<cfear_scan*> h(n); std::vector<double> poses(3 * n);
    for (size_t i = 0; i < n; i++) { h[i] = scans[i]->handle(); poses[3 * i] = Tsrc[i].t[0]; poses[3 * i + 1] = Tsrc[i].t[1]; poses[3 * i + 2] = Tsrc[i].yaw(); }
    double cov[36];
    if (soft_constraints) {  // :373-377: prior from reg_cov.back() as passed in
      double prior[36];
      for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) prior[6 * a + b] = reg_cov.back().m[a][b];
      dev_->check(cfear_register_soft(dev_->ctx(), h.data(), (int)n, poses.data(), prior, cov, &summary_), "cfear_register_soft");
    } else {
      dev_->check(cfear_register(dev_->ctx(), h.data(), (int)n, poses.data(), cov, &summary_), "cfear_register");
    }
    Tsrc.back() = Affine3d::FromXYT(poses[3 * (n - 1)], poses[3 * (n - 1) + 1], poses[3 * (n - 1) + 2]);
    if (summary_.usable) {  // :164-178: every pose passes through vectorToAffine3d(parameters), covariances get the default diagonal
      for (size_t i = 0; i + 1 < n; i++) Tsrc[i] = Affine3d::FromXYT(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]);
      for (size_t i = 0; i < n; i++) { Matrix6d m; for (int a = 0; a < 6; a++) m.m[a][a] = 0; m.m[0][0] = m.m[1][1] = 0.1 * 0.1; m.m[5][5] = 0.01 * 0.01; reg_cov[i] = m; }
      for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) reg_cov.back().m[a][b] = cov[6 * a + b];
      score_ = summary_.score;
    }
    itr_ = (size_t)summary_.outer_iterations;
    CFEAR_TIMING.Document("itrs", (double)itr_);  // n_scan_normal.cpp:161
    return summary_.success != 0;
  }
  // bool GetCost(scans, Tsrc, score, residuals) (n_scan_normal.cpp:188-213): cost and robustified residuals at the given poses
  bool GetCost(std::vector<MapNormalPtr>& scans, std::vector<Affine3d>& Tsrc, double& score, std::vector<double>& residuals) {
    const size_t n = scans.size();
    if (Tsrc.size() != n || n < 2) throw std::runtime_error("GetCost: scans/Tsrc size mismatch");  // assert at :190
    cfear_params p = dev_->params();
    p.cost = cost_ == P2L ? CFEAR_COST_P2L : (cost_ == P2D ? CFEAR_COST_P2D : CFEAR_COST_P2P); p.loss = (int)loss_; p.loss_limit = loss_limit_;
    p.weight_opt = (int)weight_opt_; p.covar_scale = cov_scale_; p.regularization = regularization_;
    dev_->set_params(p);
    std::vector<cfear_scan*> h(n); std::vector<double> poses(3 * n);
    for (size_t i = 0; i < n; i++) { h[i] = scans[i]->handle(); poses[3 * i] = Tsrc[i].t[0]; poses[3 * i + 1] = Tsrc[i].t[1]; poses[3 * i + 2] = Tsrc[i].yaw(); }
    int cap = 0; dev_->check(cfear_scan_size(dev_->ctx(), h.back(), &cap), "cfear_scan_size");
    cap = 2 * (int)(n - 1) * (cap > 0 ? cap : 1);
    residuals.assign((size_t)cap, 0.0);
    int nres = -1;
    const int rc = cfear_get_cost(dev_->ctx(), h.data(), (int)n, poses.data(), (int)itr_, &score, residuals.data(), cap, &nres);
    if (rc == CFEAR_ERR_EMPTY) { residuals.clear(); return false; }  // "too few residuals" (:205-208)
    dev_->check(rc, "cfear_get_cost");
    residuals.resize((size_t)nres);
    score_ = score / (double)(nres > 1 ? nres : 1);  // :211
    return true;
  }
  double getScore() const { return score_; }
  bool GetCovarianceScaler(double& cov_scale) const {  // n_scan_normal.cpp:435-441
    if (summary_.num_residuals - 3 == 0) return false; cov_scale = summary_.final_cost / (summary_.num_residuals - 3); return true; }
  cfear_reg_summary summary_ {};  // stands in for ceres::Solver::Summary (registration.h:110)
  size_t itr_ = 0;
 private:
  DevicePtr dev_; cost_metric cost_; loss_type loss_; double loss_limit_; weightoption weight_opt_;
  double cov_scale_ = 1, regularization_ = 0.01, score_ = 0; int max_itr_association_ = 8, max_itr_solver_ = 20;
};

// ---- OdometryKeyframeFuser (odometrykeyframefuser.h:67-260): the caller of the hot path ------------------
class OdometryKeyframeFuser {
 public:
  class Parameters {  // odometrykeyframefuser.h:72-114 (topic names and ROS-only switches omitted)
   public:
    std::string cost_type = "P2L"; weightoption weight_opt = Uniform; int submap_scan_size = 3; bool weight_intensity_ = false;
    bool use_guess = true, disable_registration = false, soft_constraint = false, compensate = true, radar_ccw = false, use_keyframe = true;
    double res = 3.5, min_keyframe_dist_ = 1.5, min_keyframe_rot_deg_ = 5; std::string loss_type_ = "Huber"; double loss_limit_ = 0.1;
    double covar_scale_ = 1.0, regularization_ = 0.0;
    // cost-sampling covariance (odometrykeyframefuser.h:104-110; the samples-to-file switch is not mirrored)
    bool estimate_cov_by_sampling = false; double cov_sampling_xy_range = 0.4, cov_sampling_yaw_range = 0.0043625;
    unsigned int cov_sampling_samples_per_axis = 3; double cov_sampling_covariance_scaler = 4.0;
  };
  OdometryKeyframeFuser(const DevicePtr& dev, const Parameters& pars, bool disable_callback = true) : dev_(dev), par(pars) {
    (void)disable_callback;
    if (!(par.res > 0.05 && par.submap_scan_size >= 1)) throw std::runtime_error("assert(par.res>0.05 && par.submap_scan_size>=1)");  // odometrykeyframefuser.cpp:25
    radar_reg.reset(new n_scan_normal_reg(dev_, Str2Cost(par.cost_type), Str2loss(par.loss_type_), par.loss_limit_, par.weight_opt));  // :27-30
    radar_reg->SetD2dPar(par.covar_scale_, par.regularization_);                                                                    // :32
  }
  // void pointcloudCallback(cloud_filtered, cloud_filtered_peaks, Tcurr, t [, cov]) (odometrykeyframefuser.cpp:397-411).
  // cloud / cloud_peaks are compensated in place like the reference does (:147-150).
  void pointcloudCallback(CloudPtr& cloud_filtered, CloudPtr& cloud_filtered_peaks, Affine3d& Tcurr, uint64_t t, Matrix6d* cov_curr = nullptr) {
    updated = false;
    processFrame(cloud_filtered, cloud_filtered_peaks, t);
    nr_callbacks_++;
    Tcurr = Tcurrent;
    if (cov_curr) *cov_curr = cov_current;
  }
  Affine3d GetCurrentPose() const { return Tcurrent; }
  size_t NumKeyframes() const { return keyframes_.size(); }
  bool updated = false;
  std::shared_ptr<n_scan_normal_reg> radar_reg;
 private:
  struct Keyframe { MapNormalPtr cloud_normal_; Affine3d pose; };
  static bool KeyFrameBasedFuse(const Affine3d& diff, bool use_keyframe, double min_keyframe_dist, double min_keyframe_rot_deg) {  // :62-73
    if (!use_keyframe) return true;
    return diff.translation_norm() > min_keyframe_dist || std::fabs(diff.yaw()) > (min_keyframe_rot_deg * M_PI / 180.0);
  }
  static bool AccelerationVelocitySanityCheck(const Affine3d& Tmot_prev, const Affine3d& Tmot_curr) {  // :76-94
    const double dt = 0.25, vel_limit = 200, acc_limit = 200;
    const double vel = Tmot_curr.translation_norm() / dt;
    const double ax = (Tmot_curr.t[0] - Tmot_prev.t[0]) / (dt * dt), ay = (Tmot_curr.t[1] - Tmot_prev.t[1]) / (dt * dt);
    return !(std::sqrt(ax * ax + ay * ay) > acc_limit) && !(vel > vel_limit);
  }
  void processFrame(CloudPtr& cloud, CloudPtr& cloud_peaks, uint64_t) {  // :143-259
    const Affine3d TprevMot(Tmot);
    DeviceCloudPtr dcloud = UploadCloud(dev_, *cloud), dpeaks = UploadCloud(dev_, *cloud_peaks);
    if (par.compensate) {  // :147-150
      Compensate(dev_, *dcloud, TprevMot, par.radar_ccw); Compensate(dev_, *dpeaks, TprevMot, par.radar_ccw);
      cloud = DownloadCloud(dev_, dcloud->h); cloud_peaks = DownloadCloud(dev_, dpeaks->h);
    }
    MapNormalPtr Pcurrent(new MapPointNormal(dev_, *dcloud, (float)par.res, Vector2d(), par.weight_intensity_, false));  // :161
    CFEAR_TIMING.Document("Surface points", (double)Pcurrent->GetSize());  // pointnormal.cpp:87
    const Affine3d Tguess = par.use_guess ? T_prev * TprevMot : T_prev;  // :164-168
    if (keyframes_.empty()) {  // :171-177
      keyframes_.push_back({Pcurrent, Affine3d::Identity()}); updated = true; return;
    }
    std::vector<Matrix6d> cov_vek; std::vector<MapNormalPtr> scans_vek; std::vector<Affine3d> T_vek;  // FormatScans :478-494
    for (auto& k : keyframes_) { cov_vek.push_back(Matrix6d()); scans_vek.push_back(k.cloud_normal_); T_vek.push_back(k.pose); }
    cov_vek.push_back(Matrix6d()); scans_vek.push_back(Pcurrent); T_vek.push_back(Tguess);
    if (!par.disable_registration) (void)radar_reg->Register(scans_vek, T_vek, cov_vek, par.soft_constraint);  // :184-186: the result lands in a shadowed variable
    Tcurrent = T_vek.back(); cov_current = cov_vek.back();  // :195-196
    const Affine3d Tmot_current = T_prev.inverse() * Tcurrent;
    if (!AccelerationVelocitySanityCheck(Tmot, Tmot_current)) Tcurrent = Tguess;  // :198-199
    Tmot = T_prev.inverse() * Tcurrent;  // :200
    if (par.estimate_cov_by_sampling) {  // :203-208
      Matrix6d cov_sampled;
      if (approximateCovarianceBySampling(scans_vek, T_vek, cov_sampled)) { cov_current = cov_sampled; cov_vek.back() = cov_sampled; }
    }
    const Affine3d Tkeydiff = keyframes_.back().pose.inverse() * Tcurrent;  // :227
    const bool fuse = KeyFrameBasedFuse(Tkeydiff, par.use_keyframe, par.min_keyframe_dist_, par.min_keyframe_rot_deg_);
    CFEAR_TIMING.Document("velocity", Tmot.translation_norm() / 0.25);  // :231
    if (fuse) {  // :234-249, AddToReference :470-476
      keyframes_.push_back({Pcurrent, Tcurrent});
      if (keyframes_.size() > (size_t)par.submap_scan_size) keyframes_.erase(keyframes_.begin());
      updated = true;
    }
    T_prev = Tcurrent;  // :257
  }
  // bool approximateCovarianceBySampling(scans_vek, T_vek, cov_sampled) (odometrykeyframefuser.cpp:261-380)
  bool approximateCovarianceBySampling(std::vector<MapNormalPtr>& scans_vek, const std::vector<Affine3d>& T_vek, Matrix6d& cov_sampled) {
    const size_t n = scans_vek.size();
    std::vector<cfear_scan*> h(n); std::vector<double> poses(3 * n);
    for (size_t i = 0; i < n; i++) { h[i] = scans_vek[i]->handle(); poses[3 * i] = T_vek[i].t[0]; poses[3 * i + 1] = T_vek[i].t[1]; poses[3 * i + 2] = T_vek[i].yaw(); }
    double cov[36]; int ok = 0;
    dev_->check(cfear_cov_by_sampling(dev_->ctx(), h.data(), (int)n, poses.data(), (int)radar_reg->itr_, par.cov_sampling_xy_range, par.cov_sampling_yaw_range,
                                      (int)par.cov_sampling_samples_per_axis, par.cov_sampling_covariance_scaler, radar_reg->summary_.final_cost,
                                      radar_reg->summary_.num_residuals, cov, &ok, nullptr), "cfear_cov_by_sampling");
    if (!ok) return false;
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) cov_sampled.m[a][b] = cov[6 * a + b];
    return true;
  }
  DevicePtr dev_; Parameters par;
  Affine3d Tcurrent, T_prev, Tmot; Matrix6d cov_current; std::vector<Keyframe> keyframes_; size_t nr_callbacks_ = 0;
};

}  // namespace CFEAR_Radarodometry
