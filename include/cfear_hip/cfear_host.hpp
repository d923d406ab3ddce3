// cfear_host.hpp -- C++ host-side mirror of the reference's interfaces for the hot path, on top of
// the C ABI (include/cfear_hip.h). Same class names, argument meaning and call structure as
//   radarDriver            (radar_driver.h:32-120,   radar_driver.cpp:23-176)
//   Compensate             (utils.h:49,               utils.cpp:96-113)
//   MapPointNormal         (pointnormal.h:110-243,   pointnormal.cpp:65-90, 238-254)
//   n_scan_normal_reg      (n_scan_normal.h:27-85,   n_scan_normal.cpp:82-187)
//   OdometryKeyframeFuser  (odometrykeyframefuser.h:67-260, odometrykeyframefuser.cpp:143-259)
// written against a handful of adapter functions over the ROS / PCL / Eigen / OpenCV types of those interfaces: POD stand-ins
// here (cfear_types_pod.hpp; none of those libraries are in this image), the real types in a tree that has them
// (include/cfear_radarodometry/*.h, the drop-in headers). Constructors have the reference's signatures: the device context
// is a process-wide default created on first use (Device::Default) unless one is passed explicitly, and every object keeps
// its own parameter snapshot, applied for the duration of its calls only (two objects with different settings on one
// device do not see each other).
// Error behaviour: where the reference prints and calls exit(0) this layer throws std::runtime_error
// carrying cfear_last_error(); bool returns are kept.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <sstream>
#include <string>
#include <vector>

#include "../cfear_hip.h"
// The ROS / PCL / Eigen / OpenCV types crossing these interfaces come from one of two headers defining the same names and
// adapter functions: include/cfear_radarodometry/cfear_types_ros.h (real types; included first by the drop-in headers of
// that directory) or, in this image, the POD stand-ins:
#ifndef CFEAR_HOST_TYPES_DEFINED
#include "cfear_types_pod.hpp"
#endif

namespace CFEAR_Radarodometry {

inline void Affine3dToVectorXYeZ(const Affine3d& T, std::vector<double>& par) { par.resize(3); par[0] = cfear_tx(T); par[1] = cfear_ty(T); par[2] = cfear_yaw(T); }  // utils.cpp:115-122
inline Affine3d vectorToAffine3d(const std::vector<double>& v) { return cfear_from_xyt(v[0], v[1], v[2]); }

typedef enum costmetric { P2P, P2L, P2D } cost_metric;                                        // registration.h:55
typedef enum losstype { None, Huber, Cauchy, SoftLOne, Combined, Tukey } loss_type;          // registration.h:60
typedef enum weight_options { Uniform = 0, Sim_N = 1, Sim_direciton = 2, Sim_scale = 3, Combined_weights = 4 } weightoption;  // :50
inline cost_metric Str2Cost(const std::string& s) { return s == "P2L" ? P2L : (s == "P2D" ? P2D : P2P); }  // registration.cpp:30-37
inline loss_type Str2loss(const std::string& s) {                                              // registration.cpp:50-65
  if (s == "Cauchy") return Cauchy; if (s == "SoftLOne") return SoftLOne; if (s == "Combined") return Combined;
  if (s == "Tukey") return Tukey; if (s == "None") return None; return Huber; }

// ---- timing (statistics.h / statistics.cpp:10-51): same stage names as the reference ------------
#ifndef CFEAR_TIMING  // (a tree with the reference's statistics.h uses its global `timing` object: cfear_types_ros.h)
struct statistics {
  std::map<std::string, std::vector<double>> executionTimes;
  void Document(const std::string& name, double value) { executionTimes[name].push_back(value); }
  std::string GetStatistics() const { std::string s; for (auto& kv : executionTimes) { double m = 0; for (double v : kv.second) m += v; m /= kv.second.empty() ? 1 : kv.second.size(); s += kv.first + ", mean " + std::to_string(m) + ", count " + std::to_string(kv.second.size()) + "\n"; } return s; }
};
inline statistics& timing_instance() { static statistics t; return t; }
#define CFEAR_TIMING CFEAR_Radarodometry::timing_instance()
#endif

// ---- device context shared by the mirrored classes (one per thread / sequence) ------------------
class Device;
typedef std::shared_ptr<Device> DevicePtr;
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
  // Process-wide default context of the reference-signature constructors: created on first use for the polar image shape
  // first seen (A x R; 400 x 3768 when a cloud arrives before any image), on HIP device CFEAR_DEVICE (environment, default 0).
  static DevicePtr& default_slot() { static DevicePtr d; return d; }
  static void SetDefault(const DevicePtr& d) { default_slot() = d; }
  static DevicePtr Default(int A = 400, int R = 3768) {
    DevicePtr& d = default_slot();
    if (!d || (A > 0 && R > 0 && (A != d->A() || R != d->R()) && !d->shape_seen_)) {
      cfear_params p; cfear_default_params(&p);
      const char* e = getenv("CFEAR_DEVICE");
      d.reset(new Device(p, A, R, e ? atoi(e) : 0));
    }
    return d;
  }
  static DevicePtr DefaultForImage(int A, int R) {  // the first image fixes the shape of the default context
    DevicePtr d = Default(A, R);
    if (d->A() != A || d->R() != R) throw std::runtime_error("polar image shape differs from the device context");
    d->shape_seen_ = true;
    return d;
  }
 private:
  cfear_ctx* ctx_ = nullptr; cfear_params par_; int A_, R_; bool shape_seen_ = false;
};
// an object's parameter snapshot applied to the context for the duration of one call
class ScopedParams {
 public:
  ScopedParams(const DevicePtr& d, const cfear_params& p) : d_(d), saved_(d->params()) { d_->set_params(p); }
  ~ScopedParams() { try { d_->set_params(saved_); } catch (...) {} }
 private:
  DevicePtr d_; cfear_params saved_;
};

// device-resident cloud handle travelling with the host cloud (keeps the data on the GPU between stages)
struct DeviceCloud { DevicePtr dev; cfear_cloud* h = nullptr; ~DeviceCloud() { if (h) cfear_cloud_release(dev->ctx(), h); } };
typedef std::shared_ptr<DeviceCloud> DeviceCloudPtr;

inline CloudPtr DownloadCloud(const DevicePtr& dev, cfear_cloud* h) {
  int n = 0; dev->check(cfear_cloud_size(dev->ctx(), h, &n), "cfear_cloud_size");
  std::vector<float> xyi(3 * (size_t)(n > 0 ? n : 1));
  dev->check(cfear_cloud_download(dev->ctx(), h, xyi.data(), n, &n), "cfear_cloud_download");
  CloudPtr c = cfear_make_cloud();
  cfear_cloud_from_xyi(*c, xyi.data(), (size_t)n);
  return c;
}
inline DeviceCloudPtr UploadCloud(const DevicePtr& dev, const PointCloudXYZI& c) {
  std::vector<float> xyi;
  cfear_cloud_to_xyi(c, xyi);
  DeviceCloudPtr d(new DeviceCloud()); d->dev = dev;
  dev->check(cfear_cloud_upload(dev->ctx(), xyi.data(), (int)cfear_cloud_size(c), &d->h), "cfear_cloud_upload");
  return d;
}

// ---- device twins of host clouds (round 6) ------------------------------------------------------------
// The reference's interfaces hand clouds over as host objects (radar_driver.h:90, utils.h:49, pointnormal.h:118), but on this
// path they are made on the device (radarDriver), changed on the device (Compensate) and consumed on the device (MapPointNormal).
// Every host cloud this layer fills is therefore remembered together with the device cloud it is a copy of: address, size and a
// checksum of its (x, y, intensity) values. A later call that receives the same object with the same content works on the device
// twin - no upload; one whose content was changed by the caller in between (or another object at a recycled address) does not
// match and takes the upload route. The host copy is always brought up to date before a call returns (unmodified caller code reads
// it: cloud->size() at offline_odometry.cpp:104, FormatScanMsg at odometrykeyframefuser.cpp:207).
inline uint64_t XyiChecksum(const float* xyi, size_t n) {
  // four independent multiply-xor lanes over 8-byte words (a single dependent chain over 10 k floats cost 10-16 us per cloud, seven times
  // per sweep: more than the kernels it guards)
  const unsigned char* p = reinterpret_cast<const unsigned char*>(xyi);
  const size_t bytes = 12 * n, words = bytes / 8;
  uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ (uint64_t)n, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
  size_t w = 0;
  for (; w + 4 <= words; w += 4) {
    uint64_t v[4]; std::memcpy(v, p + 8 * w, 32);
    for (int k = 0; k < 4; k++) { h[k] = (h[k] ^ v[k]) * 0x100000001B3ull; h[k] ^= h[k] >> 29; }
  }
  for (; w < words; w++) { uint64_t v; std::memcpy(&v, p + 8 * w, 8); h[w & 3] = (h[w & 3] ^ v) * 0x100000001B3ull; h[w & 3] ^= h[w & 3] >> 29; }
  if (bytes & 7) { uint64_t v = 0; std::memcpy(&v, p + 8 * words, bytes & 7); h[0] = (h[0] ^ v) * 0x100000001B3ull; }
  uint64_t r = h[0];
  for (int k = 1; k < 4; k++) { r = (r ^ h[k]) * 0xFF51AFD7ED558CCDull; r ^= r >> 33; }
  return r;
}
struct CloudTwin {
  const void* host = nullptr; size_t n = 0; uint64_t sum = 0;  // the host cloud object and what it held when the twin was last in step
  DeviceCloudPtr dev;
  const void* sibling = nullptr;  // the other cloud of the pair radarDriver::Process made (cloud <-> cloud_peaks)
  // Compensate(cloud) also ran on this sibling with the same motion (the fuser compensates both, odometrykeyframefuser.cpp:148-149);
  // the result waits here for the sibling's own call. Until then the device cloud is ahead of the host copy.
  bool ahead = false; double ahead_mot[3] = {0, 0, 0}; bool ahead_ccw = false; std::vector<float> ahead_xyi; size_t ahead_n = 0;
};
class CloudTwins {
 public:
  static CloudTwins& instance() { thread_local CloudTwins t; return t; }
  CloudTwin& put(const void* host, size_t n, uint64_t sum, const DeviceCloudPtr& dev, const void* sibling = nullptr) {
    drop(host);
    if (e_.size() >= 8) e_.erase(e_.begin());  // oldest first: a sweep touches two
    CloudTwin t; t.host = host; t.n = n; t.sum = sum; t.dev = dev; t.sibling = sibling;
    e_.push_back(std::move(t));
    return e_.back();
  }
  CloudTwin* by_address(const void* host) { for (auto& t : e_) if (t.host == host) return &t; return nullptr; }
  // the twin of a host cloud whose content is still what the twin was made from (scratch receives the cloud as xyi triples)
  CloudTwin* find(const PointCloudXYZI& c, std::vector<float>& scratch) {
    CloudTwin* t = by_address(&c);
    if (!t) return nullptr;
    cfear_cloud_to_xyi(c, scratch);
    if (t->n != cfear_cloud_size(c) || t->sum != XyiChecksum(scratch.data(), t->n)) { drop(&c); return nullptr; }
    return t;
  }
  void drop(const void* host) { for (size_t i = 0; i < e_.size(); i++) if (e_[i].host == host) { e_.erase(e_.begin() + (long)i); return; } }
  void clear() { e_.clear(); }
 private:
  std::vector<CloudTwin> e_;
};
// several device clouds into host clouds with one synchronisation; returns each cloud's checksum
inline void DownloadInto(const DevicePtr& dev, const std::vector<cfear_cloud*>& h, const std::vector<std::vector<float>*>& xyi, std::vector<int>& n) {
  const size_t m = h.size();
  std::vector<const cfear_cloud*> hc(h.begin(), h.end()); std::vector<float*> out(m); std::vector<int> cap(m);
  n.assign(m, 0);
  // sized by the previous answer for this slot, or a first guess: a cloud larger than the buffer is fetched again at its true size
  for (size_t i = 0; i < m; i++) { if (xyi[i]->size() < 3 * 8192) xyi[i]->resize(3 * 8192); out[i] = xyi[i]->data(); cap[i] = (int)(xyi[i]->size() / 3); }
  dev->check(cfear_clouds_download(dev->ctx(), hc.data(), (int)m, out.data(), cap.data(), n.data()), "cfear_clouds_download");
  bool again = false;
  for (size_t i = 0; i < m; i++) if (n[i] > cap[i]) { xyi[i]->resize(3 * (size_t)n[i]); out[i] = xyi[i]->data(); cap[i] = n[i]; again = true; }
  if (again) dev->check(cfear_clouds_download(dev->ctx(), hc.data(), (int)m, out.data(), cap.data(), n.data()), "cfear_clouds_download");
}

// ---- radarDriver (radar_driver.h:32-120) ----------------------------------------------------------
typedef enum filter_type { kstrong, CACFAR } filtertype;  // radar_driver.h:24
inline filtertype Str2filter(const std::string& str) { return str == "CA-CFAR" ? filtertype::CACFAR : filtertype::kstrong; }  // radar_driver.cpp:6-12
inline std::string Filter2str(const filtertype& filter) { return filter == filtertype::CACFAR ? "CA-CFAR" : "kstrong"; }       // :13-20

// ---- AzimuthCACFAR (cfar.h:31-46, cfar.cpp:27-87) -------------------------------------------------------
class AzimuthCACFAR {
 public:
  AzimuthCACFAR(int window_size = 40, double false_alarm_rate = 0.01, int nb_guard_cells = 5, double range_resolution = 0.0438,
                double static_threshold = 60.0, double min_distance = 2.5, double max_distance = 200.0)
      : window_size_(window_size), nb_guard_cells_(nb_guard_cells), false_alarm_rate_(false_alarm_rate), max_distance_(max_distance),
        range_res_((float)range_resolution), z_min_((float)static_threshold), min_distance_((float)min_distance) {}
  AzimuthCACFAR(const DevicePtr& dev, int window_size = 40, double false_alarm_rate = 0.01, int nb_guard_cells = 5, double range_resolution = 0.0438,
                double static_threshold = 60.0, double min_distance = 2.5, double max_distance = 200.0)
      : AzimuthCACFAR(window_size, false_alarm_rate, nb_guard_cells, range_resolution, static_threshold, min_distance, max_distance) { dev_ = dev; }
  // void getFilteredPointCloud(const cv_bridge::CvImagePtr&, PointCloud::Ptr& output) const (cfar.cpp:35)
  DeviceCloudPtr getFilteredPointCloud(const CvImagePtr& img, CloudPtr& output_pointcloud) const {
    const DevicePtr dev = dev_ ? dev_ : Device::DefaultForImage(cfear_cv_rows(img), cfear_cv_cols(img));
    cfear_params p = dev->params(); p.range_res = range_res_; p.z_min = z_min_; p.min_distance = min_distance_;
    ScopedParams sp(dev, p);
    DeviceCloudPtr d(new DeviceCloud()); d->dev = dev;
    dev->check(cfear_filter_cfar(dev->ctx(), cfear_cv_data(img), window_size_, nb_guard_cells_, (float)false_alarm_rate_, max_distance_, &d->h), "cfear_filter_cfar");
    std::vector<float> xyi; std::vector<int> n;
    DownloadInto(dev, {d->h}, {&xyi}, n);
    output_pointcloud = cfear_make_cloud();
    cfear_cloud_from_xyi(*output_pointcloud, xyi.data(), (size_t)n[0]);
    CloudTwins::instance().put(output_pointcloud.get(), (size_t)n[0], XyiChecksum(xyi.data(), (size_t)n[0]), d);
    return d;
  }
 private:
  DevicePtr dev_; int window_size_, nb_guard_cells_; double false_alarm_rate_, max_distance_; float range_res_, z_min_, min_distance_;
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
  radarDriver(const Parameters& pars, bool disable_callback = false) : par(pars) { (void)disable_callback; }  // radar_driver.h:86
  radarDriver(const DevicePtr& dev, const Parameters& pars, bool disable_callback = true) : dev_(dev), par(pars) { (void)disable_callback; }
  // void CallbackOffline(const sensor_msgs::ImageConstPtr&, PointCloud::Ptr& cloud, PointCloud::Ptr& cloud_peaks) (radar_driver.h:90,
  // radar_driver.cpp:163-176) -> Process() (:48-73). Oxford images are rows = azimuth x cols = range; the other datasets arrive
  // range-major and go through cv::rotate first (CallbackOffline dispatches on par.dataset like the reference, :165-170).
  void CallbackOffline(const ImageConstPtr& radar_image_polar, CloudPtr& cloud, CloudPtr& cloud_peaks) {
    if (cfear_image_null(radar_image_polar)) throw std::runtime_error("Radar image NULL");  // radar_driver.cpp:75-78
    int rows = 0, cols = 0;
    const uint8_t* raw = (par.dataset == "oxford" && par.filter_type_ == filtertype::kstrong) ? cfear_image_raw(radar_image_polar, &rows, &cols) : nullptr;
    if (raw) {
      // the message's own bytes go to the device first; cv_bridge's copy (radar_driver.cpp:104) is made while the filter runs
      LaunchKstrong(raw, rows, cols);
      const auto t0 = std::chrono::steady_clock::now();
      cv_polar_image = cfear_image_to_cv(radar_image_polar);
      CFEAR_TIMING.Document("driver: toCvCopy", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
      FinishKstrong(cloud, cloud_peaks);
      return;
    }
    const auto t0 = std::chrono::steady_clock::now();
    cv_polar_image = cfear_image_to_cv(radar_image_polar);
    CFEAR_TIMING.Document("driver: toCvCopy", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    if (par.dataset != "oxford") Rotate();
    Process(cloud, cloud_peaks);
  }
  // void Callback(const sensor_msgs::ImageConstPtr&) for the non-Oxford datasets (radar_driver.cpp:74-90): rows = range bins in;
  // cv::rotate(ROTATE_90_COUNTERCLOCKWISE) (:84) runs on the device, then Process()
  void Callback(const ImageConstPtr& range_major, CloudPtr& cloud, CloudPtr& cloud_peaks) {
    if (cfear_image_null(range_major)) throw std::runtime_error("Radar image NULL");
    cv_polar_image = cfear_image_to_cv(range_major);
    Rotate();
    Process(cloud, cloud_peaks);
  }
  // the device twin of the last `cloud` (it follows the host cloud: a Compensate of that cloud works on this object)
  DeviceCloudPtr device_cloud() const { return last_cloud_; }
  CvImagePtr cv_polar_image;  // latest radar image (radar_driver.h:92)
 private:
  DevicePtr device(int A, int R) {
    if (!dev_) dev_ = Device::DefaultForImage(A, R);
    if (A != dev_->A() || R != dev_->R()) throw std::runtime_error("polar image shape differs from the device context");
    return dev_;
  }
  void Rotate() {
    const int rows = cfear_cv_rows(cv_polar_image), cols = cfear_cv_cols(cv_polar_image);
    const DevicePtr dev = device(cols, rows);
    std::vector<uint8_t> out((size_t)rows * cols);
    dev->check(cfear_rotate_polar(dev->ctx(), cfear_cv_data(cv_polar_image), rows, cols, out.data()), "cfear_rotate_polar");
    cv_polar_image = cfear_cv_from_buffer(cols, rows, std::move(out), cv_polar_image);
  }
  void Process(CloudPtr& cloud, CloudPtr& cloud_peaks) {  // radar_driver.cpp:48-73
    const DevicePtr dev = device(cfear_cv_rows(cv_polar_image), cfear_cv_cols(cv_polar_image));
    if (par.filter_type_ == filtertype::CACFAR) {  // :52-56: max_distance 400.0, the peaks cloud stays empty
      AzimuthCACFAR filter(dev, par.window_size, par.false_alarm_rate, par.nb_guard_cells, par.range_res, par.z_min, par.min_distance, 400.0);
      last_cloud_ = filter.getFilteredPointCloud(cv_polar_image, cloud);
      last_peaks_.reset(new DeviceCloud()); last_peaks_->dev = dev;
      cloud_peaks = cfear_make_cloud();
    } else {
      LaunchKstrong(cfear_cv_data(cv_polar_image), cfear_cv_rows(cv_polar_image), cfear_cv_cols(cv_polar_image));
      FinishKstrong(cloud, cloud_peaks);
      return;
    }
    cfear_cloud_stamp_from_cv(*cloud, cv_polar_image); cfear_cloud_stamp_from_cv(*cloud_peaks, cv_polar_image);  // :66-67
  }
  // k-strongest + the two clouds (radar_driver.cpp:58-60): upload and launches ...
  void LaunchKstrong(const uint8_t* data, int rows, int cols) {
    const DevicePtr dev = device(rows, cols);
    cfear_params p = dev->params(); p.z_min = par.z_min; p.range_res = par.range_res; p.min_distance = par.min_distance; p.k_strongest = par.k_strongest;
    ScopedParams sp(dev, p);
    last_cloud_.reset(new DeviceCloud()); last_peaks_.reset(new DeviceCloud());
    last_cloud_->dev = dev; last_peaks_->dev = dev;
    const auto t0 = std::chrono::steady_clock::now();
    dev->check(cfear_filter_polar(dev->ctx(), data, &last_cloud_->h, &last_peaks_->h), "cfear_filter_polar");
    CFEAR_TIMING.Document("driver: image upload + launches", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  }
  // ... and both clouds back with one synchronisation, remembered as the host copies of their device twins
  void FinishKstrong(CloudPtr& cloud, CloudPtr& cloud_peaks) {
    const DevicePtr dev = last_cloud_->dev;
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<int> n;
    DownloadInto(dev, {last_cloud_->h, last_peaks_->h}, {&xyi_[0], &xyi_[1]}, n);
    CFEAR_TIMING.Document("driver: ... of which the wait and the copy out of the mirrors", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    cloud = cfear_make_cloud(); cloud_peaks = cfear_make_cloud();
    cfear_cloud_from_xyi(*cloud, xyi_[0].data(), (size_t)n[0]); cfear_cloud_from_xyi(*cloud_peaks, xyi_[1].data(), (size_t)n[1]);
    CloudTwins& tw = CloudTwins::instance();
    tw.put(cloud.get(), (size_t)n[0], XyiChecksum(xyi_[0].data(), (size_t)n[0]), last_cloud_, cloud_peaks.get());
    tw.put(cloud_peaks.get(), (size_t)n[1], XyiChecksum(xyi_[1].data(), (size_t)n[1]), last_peaks_, cloud.get());
    CFEAR_TIMING.Document("driver: wait + two clouds to the host", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    cfear_cloud_stamp_from_cv(*cloud, cv_polar_image); cfear_cloud_stamp_from_cv(*cloud_peaks, cv_polar_image);  // :66-67
  }
  DevicePtr dev_; Parameters par; DeviceCloudPtr last_cloud_, last_peaks_; std::vector<float> xyi_[2];
};

// ---- Compensate (utils.h:47-49, utils.cpp:96-113) -----------------------------------------------------
inline void Compensate(const DevicePtr& dev, DeviceCloud& cloud, const Affine3d& Tmotion, bool ccw) {
  const double mot[3] = {cfear_tx(Tmotion), cfear_ty(Tmotion), cfear_yaw(Tmotion)};
  dev->check(cfear_compensate(dev->ctx(), cloud.h, mot, ccw ? 1 : 0), "cfear_compensate");
}
// void Compensate(pcl::PointCloud<pcl::PointXYZI>& cloud, const Eigen::Affine3d& Tmotion, bool ccw) (utils.h:49): in place.
// A cloud this layer made (radarDriver, an earlier Compensate) and nobody changed since is compensated on its device twin; when it
// has a sibling from the same sweep (cloud <-> cloud_peaks) the sibling is compensated by the same motion in the same launch
// sequence and its result parked, so that the fuser's two calls (odometrykeyframefuser.cpp:148-149) cost one synchronisation.
// Anything else is uploaded first. `fallback` = the device for clouds without a twin (null: the process-wide default).
inline void CompensateOn(const DevicePtr& fallback, PointCloudXYZI& cloud, const Affine3d& Tmotion, bool ccw) {
  if (cfear_cloud_size(cloud) == 0) return;
  CloudTwins& tw = CloudTwins::instance();
  static thread_local std::vector<float> xyi, sib_xyi;
  const double mot[3] = {cfear_tx(Tmotion), cfear_ty(Tmotion), cfear_yaw(Tmotion)};
  CloudTwin* t = tw.find(cloud, xyi);
  if (t && t->ahead) {  // the sibling's call compensated this cloud already
    if (t->ahead_mot[0] == mot[0] && t->ahead_mot[1] == mot[1] && t->ahead_mot[2] == mot[2] && t->ahead_ccw == ccw) {
      cfear_cloud_from_xyi(cloud, t->ahead_xyi.data(), t->ahead_n);
      t->n = t->ahead_n; t->sum = XyiChecksum(t->ahead_xyi.data(), t->ahead_n); t->ahead = false;
      return;
    }
    tw.drop(&cloud); t = nullptr;  // another motion: the device twin is of no use any more
  }
  if (!t) {  // (xyi holds the cloud when find() looked at it; a cloud without any entry is converted here)
    const DevicePtr dev = fallback ? fallback : Device::Default();
    cfear_cloud_to_xyi(cloud, xyi);
    DeviceCloudPtr d(new DeviceCloud()); d->dev = dev;
    dev->check(cfear_cloud_upload(dev->ctx(), xyi.data(), (int)cfear_cloud_size(cloud), &d->h), "cfear_cloud_upload");
    t = &tw.put(&cloud, cfear_cloud_size(cloud), 0, d);
  }
  const DevicePtr dev = t->dev->dev;
  CloudTwin* sib = t->sibling ? tw.by_address(t->sibling) : nullptr;
  if (sib && (sib->ahead || sib->dev->dev != dev || sib->n == 0)) sib = nullptr;
  std::vector<int> n;
  if (sib) {
    dev->check(cfear_compensate_pair(dev->ctx(), t->dev->h, sib->dev->h, mot, ccw ? 1 : 0), "cfear_compensate_pair");  // one launch for the sweep's two clouds
    DownloadInto(dev, {t->dev->h, sib->dev->h}, {&xyi, &sib->ahead_xyi}, n);
    sib->ahead = true; sib->ahead_n = (size_t)n[1]; sib->ahead_ccw = ccw;
    for (int i = 0; i < 3; i++) sib->ahead_mot[i] = mot[i];
  } else {
    Compensate(dev, *t->dev, Tmotion, ccw);
    DownloadInto(dev, {t->dev->h}, {&xyi}, n);
  }
  cfear_cloud_from_xyi(cloud, xyi.data(), (size_t)n[0]);
  t->n = (size_t)n[0]; t->sum = XyiChecksum(xyi.data(), t->n);
}
inline void Compensate(PointCloudXYZI& cloud, const Affine3d& Tmotion, bool ccw) { CompensateOn(DevicePtr(), cloud, Tmotion, ccw); }
inline void Compensate(PointCloudXYZI& cloud, const std::vector<double>& mot, bool ccw) { Compensate(cloud, vectorToAffine3d(mot), ccw); }  // utils.h:47

// ---- cell / MapPointNormal (pointnormal.h:45-243) -----------------------------------------------------
struct cell {
  Vector2d u_; Matrix2d cov_; double scale_ = 0; Vector2d snormal_, orth_normal; double lambda_min = 0, lambda_max = 0;
  double sum_intensity_ = 0, avg_intensity_ = 0; size_t Nsamples_ = 0; bool valid_ = false;
  double GetPlanarity() const { return scale_; }
  // static cell GetIdentityCell(u, intensity) (pointnormal.h:58, private ctor :80-82); cov_ keeps its default Identity * 0.1 (:63)
  static cell GetIdentityCell(const Vector2d& u, const double intensity) {
    (void)intensity;
    cell c; c.u_ = u; c.cov_ = cfear_mat2(0.1, 0, 0, 0.1); c.scale_ = 1.0; c.snormal_ = cfear_vec2(1, 0); c.orth_normal = cfear_vec2(0, 1);
    c.lambda_min = 1; c.lambda_max = 1; c.sum_intensity_ = 1.0; c.avg_intensity_ = 1.0; c.Nsamples_ = 1; c.valid_ = true;
    return c;
  }
#ifdef CFEAR_HOST_BOOST_SERIALIZATION  // pointnormal.h:86-101: the same archive layout, so the reference's SaveSimpleGraph / LoadSimpleGraph
  template <class Archive>             // (types.cpp:103-130, the .sgh export for TBV-SLAM) keep working on these cells
  void serialize(Archive& ar, const unsigned int) {
    ar & u_; ar & cov_; ar & scale_; ar & snormal_; ar & lambda_min; ar & lambda_max; ar & sum_intensity_; ar & avg_intensity_; ar & Nsamples_; ar & valid_;
  }
#endif
  // cell TransformCopy(T) (pointnormal.cpp:515-527), arithmetic as written there: C = R * T * cov * R^T with the affine T
  // applied to the columns of cov (so the translation enters the product)
  cell TransformCopy(const Affine3d& T) const {
    double R[2][2]; cfear_linear2(T, R);
    const double t[2] = {cfear_tx(T), cfear_ty(T)};
    cell ct = *this;
    double M[2][2], RT[2][2], rt[2];  // R*T: linear R*R, translation R*t
    for (int i = 0; i < 2; i++) { for (int j = 0; j < 2; j++) RT[i][j] = R[i][0] * R[0][j] + R[i][1] * R[1][j]; rt[i] = R[i][0] * t[0] + R[i][1] * t[1]; }
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) M[i][j] = RT[i][0] * cov_(0, j) + RT[i][1] * cov_(1, j) + rt[i];
    ct.cov_ = cfear_mat2(M[0][0] * R[0][0] + M[0][1] * R[0][1], M[0][0] * R[1][0] + M[0][1] * R[1][1],
                         M[1][0] * R[0][0] + M[1][1] * R[0][1], M[1][0] * R[1][0] + M[1][1] * R[1][1]);
    ct.snormal_ = cfear_vec2(R[0][0] * snormal_(0) + R[0][1] * snormal_(1), R[1][0] * snormal_(0) + R[1][1] * snormal_(1));
    ct.orth_normal = cfear_vec2(R[0][0] * orth_normal(0) + R[0][1] * orth_normal(1), R[1][0] * orth_normal(0) + R[1][1] * orth_normal(1));
    ct.u_ = cfear_vec2(R[0][0] * u_(0) + R[0][1] * u_(1) + t[0], R[1][0] * u_(0) + R[1][1] * u_(1) + t[1]);
    return ct;
  }
};
class MapPointNormal;
typedef CFEAR_SHARED_PTR<MapPointNormal> MapNormalPtr;
class MapPointNormal {
 public:
  // MapPointNormal(cld, radius, origin = (0,0), weight_intensity = false, raw = false) (pointnormal.h:118, pointnormal.cpp:65-90)
  // A cloud that still equals its device twin (radarDriver / Compensate made it, nobody changed it) is not uploaded again.
  MapPointNormal(const CloudPtr& cld, float radius, const Vector2d& origin = Vector2d(0, 0), const bool weight_intensity = false, const bool raw = false)
      : input_(cld) { Init(DevicePtr(), radius, origin, weight_intensity, raw); }
  // ... with the device to use when the cloud has no twin (the reference-signature constructor takes the process-wide default)
  MapPointNormal(const DevicePtr& dev, const CloudPtr& cld, float radius, const Vector2d& origin = Vector2d(0, 0), const bool weight_intensity = false, const bool raw = false)
      : input_(cld) { Init(dev, radius, origin, weight_intensity, raw); }
  // MapPointNormal(cld, radius, cell_orig, T) (pointnormal.h:120, pointnormal.cpp:91-110): transformed copy of existing cells
  MapPointNormal(const CloudPtr& cld, float radius, std::vector<cell>& cell_orig, const Affine3d& T) : dev_(Device::Default()), input_(cld) {
    (void)radius;
    std::vector<cell> cs;
    for (const cell& c : cell_orig) cs.push_back(c.TransformCopy(T));
    FromCells(cs);
  }
  // the same from a cloud that is already on the device (radarDriver::device_cloud(): no upload)
  MapPointNormal(const DevicePtr& dev, const DeviceCloud& cld, float radius, const Vector2d& origin = Vector2d(0, 0), bool weight_intensity = false, bool raw = false) : dev_(dev) {
    if (raw) { CloudPtr host = DownloadCloud(dev_, cld.h); if (cfear_cloud_size(*host) == 0) throw std::runtime_error("error, cloud empty"); BuildRaw(*host); return; }
    Build(cld, radius, origin, weight_intensity);
  }
  ~MapPointNormal() { if (scan_) cfear_scan_release(dev_->ctx(), scan_); }
  MapPointNormal(const MapPointNormal&) = delete;
  // RViz marker publisher of the reference (pointnormal.h:235, pointnormal.cpp:299-...): visualisation only, off the path - kept
  // as a no-op so that odometrykeyframefuser.cpp:210 compiles unchanged
  static void PublishMap(const std::string& topic, CFEAR_SHARED_PTR<MapPointNormal> map, const Affine3d& T, const std::string& frame_id,
                         const int value = 0, float alpha = 1.0) { (void)topic; (void)map; (void)T; (void)frame_id; (void)value; (void)alpha; }
  size_t GetSize() { int n = 0; dev_->check(cfear_scan_size(dev_->ctx(), scan_, &n), "cfear_scan_size"); return (size_t)n; }
  std::vector<cell> GetCells() { fetch(); return cells_; }
  cell& GetCell(const size_t i) { fetch(); return cells_[i]; }
  Vector2d GetMean2d(const size_t i) { fetch(); return cells_[i].u_; }
  Matrix2d GetCov2d(const size_t i) { fetch(); return cells_[i].cov_; }
  Vector2d GetNormal2d(const size_t i) { fetch(); return cells_[i].snormal_; }
  CloudPtr GetScan() { return input_; }  // pointnormal.h:172 (null when built from a device cloud)
  std::vector<int> GetClosestIdx(const Vector2d& p, double d) {  // pointnormal.cpp:238-254
    const double q[2] = {p(0), p(1)}; int32_t idx = -1;
    dev_->check(cfear_scan_closest(dev_->ctx(), scan_, q, 1, d, &idx), "cfear_scan_closest");
    return idx >= 0 ? std::vector<int>{idx} : std::vector<int>();
  }
  std::vector<cell*> GetClosest(Vector2d& p, double d) {  // pointnormal.cpp:225-236
    fetch();
    std::vector<cell*> out;
    for (int i : GetClosestIdx(p, d)) out.push_back(&cells_[(size_t)i]);
    return out;
  }
  // double GetCellRelTimeStamp(index, ccw) (pointnormal.cpp:139-143) with GetRelTimeStamp (utils.h:28-32)
  double GetCellRelTimeStamp(const size_t index, const bool ccw) {
    fetch();
    const double a = std::atan2(cells_[index].u_(1), cells_[index].u_(0));
    const double d = (a > 0.00001 ? a : (2 * M_PI + a)) / (2 * M_PI);
    return ccw ? -(d - 0.5) : (d - 0.5);
  }
  // std::vector<cell> TransformCells(T) (pointnormal.cpp:352-361) -> cell::TransformCopy (:515-527)
  std::vector<cell> TransformCells(const Affine3d& T) {
    fetch();
    std::vector<cell> out;
    for (const cell& c : cells_) out.push_back(c.TransformCopy(T));
    return out;
  }
  // MapNormalPtr TransformMap(T) (pointnormal.cpp:135-137): new map of transformed cells
  MapNormalPtr TransformMap(const Affine3d& T) { fetch(); return MapNormalPtr(new MapPointNormal(input_, 0.f, cells_, T)); }
  cfear_scan* handle() const { return scan_; }
  const DevicePtr& device() const { return dev_; }
  static double downsample_factor;  // pointnormal.h:241 (read when a map is built)
#ifdef CFEAR_HOST_BOOST_SERIALIZATION
  // pointnormal.h:201-226: cells, input_, downsampled_, radius_, weight_intensity_ in this order; loading rebuilds the device
  // scan from the cells (the reference rebuilds its kd-tree, :225)
  MapPointNormal() : dev_(Device::Default()) {}
  template <class Archive>
  void save(Archive& ar, const unsigned int) const {
    const_cast<MapPointNormal*>(this)->fetch();
    CFEAR_DOWNSAMPLED_PTR downsampled = cfear_make_downsampled(cells_);
    ar & cells_; ar & input_; ar & downsampled; ar & radius_; ar & weight_intensity_;
  }
  template <class Archive>
  void load(Archive& ar, const unsigned int) {
    std::vector<cell> cs; CFEAR_DOWNSAMPLED_PTR downsampled;
    ar & cs; ar & input_; ar & downsampled; ar & radius_; ar & weight_intensity_;
    if (scan_) { cfear_scan_release(dev_->ctx(), scan_); scan_ = nullptr; }
    FromCells(cs);
  }
  BOOST_SERIALIZATION_SPLIT_MEMBER()
#endif
 private:
  void Init(const DevicePtr& fallback, float radius, const Vector2d& origin, bool weight_intensity, bool raw) {
    const CloudPtr& cld = input_;
    if (!cld || cfear_cloud_size(*cld) == 0) throw std::runtime_error("error, cloud empty");  // pointnormal.cpp:72-75 (exit(0) there)
    static thread_local std::vector<float> xyi;
    CloudTwin* t = raw ? nullptr : CloudTwins::instance().find(*cld, xyi);
    if (t && t->ahead) t = nullptr;  // (its device cloud was compensated ahead of the host copy: not this content)
    dev_ = t ? t->dev->dev : (fallback ? fallback : Device::Default());
    if (raw) { BuildRaw(*cld); return; }
    if (t) { Build(*t->dev, radius, origin, weight_intensity); return; }
    DeviceCloudPtr d = UploadCloud(dev_, *cld);
    Build(*d, radius, origin, weight_intensity);
  }
  void Build(const DeviceCloud& cld, float radius, const Vector2d& origin, bool weight_intensity) {
    radius_ = radius; weight_intensity_ = weight_intensity;
    if (origin(0) != 0 || origin(1) != 0) throw std::runtime_error("MapPointNormal: origin must be (0,0) as in odometrykeyframefuser.cpp:161");
    cfear_params p = dev_->params(); p.res = radius; p.weight_intensity = weight_intensity ? 1 : 0; p.downsample_factor = downsample_factor;
    ScopedParams sp(dev_, p);  // this map's settings, for this call only
    const int rc = cfear_scan_create(dev_->ctx(), cld.h, &scan_);
    if (rc == CFEAR_ERR_EMPTY) throw std::runtime_error("error, cloud empty");  // pointnormal.cpp:72-75 (exit(0) there)
    dev_->check(rc, "cfear_scan_create");
  }
  void BuildRaw(const PointCloudXYZI& cld) {  // pointnormal.cpp:76-82: one identity cell per point
    std::vector<float> xyi; cfear_cloud_to_xyi(cld, xyi);
    std::vector<cell> cs;
    for (size_t i = 0; i < cfear_cloud_size(cld); i++) cs.push_back(cell::GetIdentityCell(cfear_vec2(xyi[3 * i], xyi[3 * i + 1]), xyi[3 * i + 2]));
    FromCells(cs);
  }
  void FromCells(const std::vector<cell>& cs) {
    std::vector<cfear_cell> raw(cs.size() ? cs.size() : 1);
    for (size_t i = 0; i < cs.size(); i++) {
      const cell& c = cs[i]; cfear_cell& r = raw[i];
      r.mean[0] = c.u_(0); r.mean[1] = c.u_(1); r.cov[0] = c.cov_(0, 0); r.cov[1] = c.cov_(0, 1); r.cov[2] = c.cov_(1, 1);
      r.normal[0] = c.snormal_(0); r.normal[1] = c.snormal_(1); r.orth[0] = c.orth_normal(0); r.orth[1] = c.orth_normal(1);
      r.lambda_min = c.lambda_min; r.lambda_max = c.lambda_max; r.scale = c.scale_; r.sum_intensity = c.sum_intensity_; r.avg_intensity = c.avg_intensity_;
      r.nsamples = (int32_t)c.Nsamples_; r.valid = c.valid_ ? 1 : 0;
    }
    const int rc = cfear_scan_from_cells(dev_->ctx(), raw.data(), (int)cs.size(), &scan_);
    if (rc == CFEAR_ERR_EMPTY) throw std::runtime_error("error, cloud empty");
    dev_->check(rc, "cfear_scan_from_cells");
    cells_ = cs; fetched_ = true;
  }
  void fetch() {
    if (fetched_) return;
    int n = 0; dev_->check(cfear_scan_size(dev_->ctx(), scan_, &n), "cfear_scan_size");
    std::vector<cfear_cell> raw((size_t)(n > 0 ? n : 1));
    dev_->check(cfear_scan_download_cells(dev_->ctx(), scan_, raw.data(), n, &n), "cfear_scan_download_cells");
    cells_.resize(n);
    for (int i = 0; i < n; i++) {
      cell& c = cells_[i]; const cfear_cell& r = raw[i];
      c.u_ = cfear_vec2(r.mean[0], r.mean[1]); c.cov_ = cfear_mat2(r.cov[0], r.cov[1], r.cov[1], r.cov[2]);
      c.snormal_ = cfear_vec2(r.normal[0], r.normal[1]); c.orth_normal = cfear_vec2(r.orth[0], r.orth[1]);
      c.lambda_min = r.lambda_min; c.lambda_max = r.lambda_max; c.scale_ = r.scale; c.sum_intensity_ = r.sum_intensity; c.avg_intensity_ = r.avg_intensity;
      c.Nsamples_ = (size_t)r.nsamples; c.valid_ = r.valid != 0;
    }
    fetched_ = true;
  }
  DevicePtr dev_; CloudPtr input_; cfear_scan* scan_ = nullptr; std::vector<cell> cells_; bool fetched_ = false;
  float radius_ = 0; bool weight_intensity_ = false;  // pointnormal.h:199-200 (kept for the archive)
};
inline double MapPointNormal::downsample_factor = 1;

// What reference code reads of ceres::Solver::Summary through n_scan_normal_reg::summary_ (registration.h:110):
// final_cost, num_residuals, iterations, and - at odometrykeyframefuser.cpp:191 - FullReport() for the failure message.
struct RegSummary : cfear_reg_summary {
  RegSummary() : cfear_reg_summary() {}
  bool IsSolutionUsable() const { return usable != 0; }
  std::string BriefReport() const {
    std::ostringstream o;
    o << "CFEAR-HIP report: outer iterations " << outer_iterations << ", residuals " << num_residuals << " in " << num_residual_blocks
      << " blocks, final cost " << final_cost << ", " << (usable ? "usable" : "NOT usable");
    return o.str();
  }
  std::string FullReport() const {
    std::ostringstream o;
    o << "\n" << BriefReport() << "\nouter  inner  termination  cost  pose\n";
    for (int i = 0; i < outer_iterations - 1 && i < CFEAR_MAX_OUTER; i++)
      o << i + 1 << "  " << inner_iterations[i] << "  " << (termination[i] == 0 ? "CONVERGENCE" : (termination[i] == 1 ? "NO_CONVERGENCE" : "FAILURE")) << "  "
        << outer_cost[i] << "  (" << outer_pose[i][0] << ", " << outer_pose[i][1] << ", " << outer_pose[i][2] << ")\n";
    return o.str();
  }
};

// ---- n_scan_normal_reg (n_scan_normal.h:27-85) ----------------------------------------------------------
class n_scan_normal_reg {
 public:
  n_scan_normal_reg() {}                                                                                   // n_scan_normal.h:33
  n_scan_normal_reg(const cost_metric& cost, loss_type loss = Huber, double loss_limit = 0.1, const weightoption opt = weightoption::Uniform)  // :35
      : cost_(cost), loss_(loss), loss_limit_(loss_limit), weight_opt_(opt) {}
  n_scan_normal_reg(const DevicePtr& dev, const cost_metric& cost, loss_type loss = Huber, double loss_limit = 0.1, const weightoption opt = Uniform)
      : dev_(dev), cost_(cost), loss_(loss), loss_limit_(loss_limit), weight_opt_(opt) {}
  void SetD2dPar(const double cov_scale, const double regularization) { cov_scale_ = cov_scale; regularization_ = regularization; }  // n_scan_normal.h:53
  void SetParameters(unsigned int max_itr_association, unsigned int max_itr_solver) { max_itr_association_ = (int)max_itr_association; max_itr_solver_ = (int)max_itr_solver; }
  // bool Register(scans, Tsrc, reg_cov, soft_constraints = false) (n_scan_normal.cpp:82-187); only Tsrc.back() is free.
  bool Register(std::vector<MapNormalPtr>& scans, std::vector<Affine3d>& Tsrc, std::vector<Matrix6d>& reg_cov, bool soft_constraints = false) {
    const size_t n = scans.size();
    if (Tsrc.size() != n || reg_cov.size() != n || n < 2) throw std::runtime_error("Register: scans/Tsrc/reg_cov size mismatch");  // assert at n_scan_normal.cpp:84
    const DevicePtr dev = device(scans);
    ScopedParams sp(dev, my_params(dev));  // this object's cost / loss / weights, for this call only
    std::vector<cfear_scan*> h(n); std::vector<double> poses(3 * n);
    for (size_t i = 0; i < n; i++) { h[i] = scans[i]->handle(); poses[3 * i] = cfear_tx(Tsrc[i]); poses[3 * i + 1] = cfear_ty(Tsrc[i]); poses[3 * i + 2] = cfear_yaw(Tsrc[i]); }
    double cov[36] = {0};  // (in/out at the C ABI: left as passed when no usable solution comes back)
    if (soft_constraints) {  // :373-377: prior from reg_cov.back() as passed in
      double prior[36];
      for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) prior[6 * a + b] = reg_cov.back()(a, b);
      dev->check(cfear_register_soft(dev->ctx(), h.data(), (int)n, poses.data(), prior, cov, &summary_), "cfear_register_soft");
    } else {
      dev->check(cfear_register(dev->ctx(), h.data(), (int)n, poses.data(), cov, &summary_), "cfear_register");
    }
    Tsrc.back() = cfear_from_xyt(poses[3 * (n - 1)], poses[3 * (n - 1) + 1], poses[3 * (n - 1) + 2]);
    if (summary_.usable) {  // :164-178: every pose passes through vectorToAffine3d(parameters), covariances get the default diagonal
      for (size_t i = 0; i + 1 < n; i++) Tsrc[i] = cfear_from_xyt(poses[3 * i], poses[3 * i + 1], poses[3 * i + 2]);
      for (size_t i = 0; i < n; i++) { Matrix6d m = cfear_mat6_identity(); for (int a = 0; a < 6; a++) m(a, a) = 0; m(0, 0) = m(1, 1) = 0.1 * 0.1; m(5, 5) = 0.01 * 0.01; reg_cov[i] = m; }
      for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) reg_cov.back()(a, b) = cov[6 * a + b];
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
    const DevicePtr dev = device(scans);
    ScopedParams sp(dev, my_params(dev));
    std::vector<cfear_scan*> h(n); std::vector<double> poses(3 * n);
    for (size_t i = 0; i < n; i++) { h[i] = scans[i]->handle(); poses[3 * i] = cfear_tx(Tsrc[i]); poses[3 * i + 1] = cfear_ty(Tsrc[i]); poses[3 * i + 2] = cfear_yaw(Tsrc[i]); }
    int cap = 0; dev->check(cfear_scan_size(dev->ctx(), h.back(), &cap), "cfear_scan_size");
    cap = 2 * (int)(n - 1) * (cap > 0 ? cap : 1);
    residuals.assign((size_t)cap, 0.0);
    int nres = -1;
    const int rc = cfear_get_cost(dev->ctx(), h.data(), (int)n, poses.data(), (int)itr_, &score, residuals.data(), cap, &nres);
    if (rc == CFEAR_ERR_EMPTY) { residuals.clear(); return false; }  // "too few residuals" (:205-208)
    dev->check(rc, "cfear_get_cost");
    residuals.resize((size_t)nres);
    score_ = score / (double)(nres > 1 ? nres : 1);  // :211
    return true;
  }
  double getScore() { return score_; }
  void getScore(double& score, int& num_residuals) { score = score_; num_residuals = summary_.num_residuals; }  // n_scan_normal.h:51
  bool GetCovarianceScaler(double& cov_scale) {  // n_scan_normal.cpp:435-441
    if (summary_.num_residuals - 3 == 0) return false; cov_scale = summary_.final_cost / (summary_.num_residuals - 3); return true; }
  RegSummary summary_;  // stands in for ceres::Solver::Summary (registration.h:110): a cfear_reg_summary with FullReport()
  size_t itr_ = 0;
  // device and parameter snapshot of this object (cfear_cov_by_sampling runs under them as well)
  DevicePtr device(const std::vector<MapNormalPtr>& scans) { if (!dev_) dev_ = scans.empty() ? Device::Default() : scans.back()->device(); return dev_; }
  cfear_params my_params(const DevicePtr& dev) const {
    cfear_params p = dev->params();
    p.cost = cost_ == P2L ? CFEAR_COST_P2L : (cost_ == P2D ? CFEAR_COST_P2D : CFEAR_COST_P2P); p.loss = (int)loss_; p.loss_limit = loss_limit_;
    p.weight_opt = (int)weight_opt_; p.covar_scale = cov_scale_; p.regularization = regularization_;
    p.max_itr_association = max_itr_association_; p.max_solver_iterations = max_itr_solver_;
    return p;
  }
 private:
  DevicePtr dev_; cost_metric cost_ = P2L; loss_type loss_ = Huber; double loss_limit_ = 0.1; weightoption weight_opt_ = weightoption::Uniform;  // registration.h:117-120
  double cov_scale_ = 1, regularization_ = 0.01, score_ = 0; int max_itr_association_ = 8, max_itr_solver_ = 20;
};

// ---- OdometryKeyframeFuser (odometrykeyframefuser.h:67-260): the caller of the hot path ------------------
class OdometryKeyframeFuser {
 public:
  class Parameters {  // odometrykeyframefuser.h:72-114 (topic names and ROS-only switches omitted)
   public:
    std::string cost_type = "P2L"; weightoption weight_opt = Uniform; int submap_scan_size = 3; bool weight_intensity_ = false;
    bool use_guess = true, disable_registration = false, soft_constraint = false, compensate = true, radar_ccw = false, use_keyframe = true;
    bool use_raw_pointcloud = false;  // odometrykeyframefuser.h:96
    double res = 3.5, min_keyframe_dist_ = 1.5, min_keyframe_rot_deg_ = 5; std::string loss_type_ = "Huber"; double loss_limit_ = 0.1;
    double covar_scale_ = 1.0, regularization_ = 0.0;
    // cost-sampling covariance (odometrykeyframefuser.h:104-110; the samples-to-file switch is not mirrored)
    bool estimate_cov_by_sampling = false; double cov_sampling_xy_range = 0.4, cov_sampling_yaw_range = 0.0043625;
    unsigned int cov_sampling_samples_per_axis = 3; double cov_sampling_covariance_scaler = 4.0;
  };
  OdometryKeyframeFuser(const Parameters& pars, bool disable_callback = false) : par(pars) { Init(disable_callback); }  // odometrykeyframefuser.h:187
  OdometryKeyframeFuser(const DevicePtr& dev, const Parameters& pars, bool disable_callback = true) : dev_(dev), par(pars) { Init(disable_callback); }
  // void pointcloudCallback(cloud_filtered, cloud_filtered_peaks, Tcurr, t [, cov]) (odometrykeyframefuser.cpp:397-411).
  // cloud / cloud_peaks are compensated in place like the reference does (:147-150).
  void pointcloudCallback(CloudPtr& cloud_filtered, CloudPtr& cloud_filtered_peaks, Affine3d& Tcurr, uint64_t t, Matrix6d* cov_curr = nullptr) {
    updated = false;
    processFrame(cloud_filtered, cloud_filtered_peaks, t);
    nr_callbacks_++;
    Tcurr = Tcurrent;
    if (cov_curr) *cov_curr = cov_current;
  }
  void pointcloudCallback(CloudPtr& cloud_filtered, CloudPtr& cloud_filtered_peaks, Affine3d& Tcurr, uint64_t t, Matrix6d& cov_curr) {  // :409-411
    pointcloudCallback(cloud_filtered, cloud_filtered_peaks, Tcurr, t, &cov_curr);
  }
  Affine3d GetCurrentPose() const { return Tcurrent; }
  size_t NumKeyframes() const { return keyframes_.size(); }
  bool updated = false;
  std::shared_ptr<n_scan_normal_reg> radar_reg;
 private:
  struct Keyframe { MapNormalPtr cloud_normal_; Affine3d pose; };
  void Init(bool disable_callback) {
    (void)disable_callback;
    if (!(par.res > 0.05 && par.submap_scan_size >= 1)) throw std::runtime_error("assert(par.res>0.05 && par.submap_scan_size>=1)");  // odometrykeyframefuser.cpp:25
    if (!dev_) dev_ = Device::Default();
    radar_reg.reset(new n_scan_normal_reg(dev_, Str2Cost(par.cost_type), Str2loss(par.loss_type_), par.loss_limit_, par.weight_opt));  // :27-30
    radar_reg->SetD2dPar(par.covar_scale_, par.regularization_);                                                                    // :32
  }
  static bool KeyFrameBasedFuse(const Affine3d& diff, bool use_keyframe, double min_keyframe_dist, double min_keyframe_rot_deg) {  // :62-73
    if (!use_keyframe) return true;
    return cfear_tnorm(diff) > min_keyframe_dist || std::fabs(cfear_yaw(diff)) > (min_keyframe_rot_deg * M_PI / 180.0);
  }
  static bool AccelerationVelocitySanityCheck(const Affine3d& Tmot_prev, const Affine3d& Tmot_curr) {  // :76-94
    const double dt = 0.25, vel_limit = 200, acc_limit = 200;
    const double vel = cfear_tnorm(Tmot_curr) / dt;
    const double ax = (cfear_tx(Tmot_curr) - cfear_tx(Tmot_prev)) / (dt * dt), ay = (cfear_ty(Tmot_curr) - cfear_ty(Tmot_prev)) / (dt * dt);
    return !(std::sqrt(ax * ax + ay * ay) > acc_limit) && !(vel > vel_limit);
  }
  void processFrame(CloudPtr& cloud, CloudPtr& cloud_peaks, uint64_t) {  // :143-259
    // The same calls on the same host objects as the reference makes (Compensate(*cloud, ...) twice, then MapPointNormal(cloud, ...)):
    // clouds that came from radarDriver::CallbackOffline on this thread are found as device twins and never uploaded; dev_ is
    // only the device for clouds from elsewhere.
    const Affine3d TprevMot(Tmot);
    const auto t0 = std::chrono::steady_clock::now();
    if (par.compensate) {  // :147-150
      CompensateOn(dev_, *cloud, TprevMot, par.radar_ccw);
      CompensateOn(dev_, *cloud_peaks, TprevMot, par.radar_ccw);
    }
    const auto t1 = std::chrono::steady_clock::now();
    MapNormalPtr Pcurrent(new MapPointNormal(dev_, cloud, (float)par.res, Vector2d(0, 0), par.weight_intensity_, par.use_raw_pointcloud));  // :161
    CFEAR_TIMING.Document("Surface points", (double)Pcurrent->GetSize());  // pointnormal.cpp:87
    const auto t2 = std::chrono::steady_clock::now();
    CFEAR_TIMING.Document("compensate", std::chrono::duration<double, std::milli>(t1 - t0).count());     // odometrykeyframefuser.cpp:253
    CFEAR_TIMING.Document("build_normals", std::chrono::duration<double, std::milli>(t2 - t1).count());  // :254
    const Affine3d Tguess = par.use_guess ? T_prev * TprevMot : T_prev;  // :164-168
    if (keyframes_.empty()) {  // :171-177
      keyframes_.push_back({Pcurrent, cfear_from_xyt(0, 0, 0)}); updated = true; return;
    }
    std::vector<Matrix6d> cov_vek; std::vector<MapNormalPtr> scans_vek; std::vector<Affine3d> T_vek;  // FormatScans :478-494
    for (auto& k : keyframes_) { cov_vek.push_back(cfear_mat6_identity()); scans_vek.push_back(k.cloud_normal_); T_vek.push_back(k.pose); }
    cov_vek.push_back(cfear_mat6_identity()); scans_vek.push_back(Pcurrent); T_vek.push_back(Tguess);
    if (!par.disable_registration) (void)radar_reg->Register(scans_vek, T_vek, cov_vek, par.soft_constraint);  // :184-186: the result lands in a shadowed variable
    CFEAR_TIMING.Document("register", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t2).count());  // :255
    Tcurrent = T_vek.back(); cov_current = cov_vek.back();  // :195-196
    const Affine3d Tmot_current = T_prev.inverse() * Tcurrent;
    if (!AccelerationVelocitySanityCheck(Tmot, Tmot_current)) Tcurrent = Tguess;  // :198-199
    Tmot = T_prev.inverse() * Tcurrent;  // :200
    if (par.estimate_cov_by_sampling) {  // :203-208
      Matrix6d cov_sampled = cfear_mat6_identity();
      if (approximateCovarianceBySampling(scans_vek, T_vek, cov_sampled)) { cov_current = cov_sampled; cov_vek.back() = cov_sampled; }
    }
    const Affine3d Tkeydiff = keyframes_.back().pose.inverse() * Tcurrent;  // :227
    const bool fuse = KeyFrameBasedFuse(Tkeydiff, par.use_keyframe, par.min_keyframe_dist_, par.min_keyframe_rot_deg_);
    CFEAR_TIMING.Document("velocity", cfear_tnorm(Tmot) / 0.25);  // :231
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
    for (size_t i = 0; i < n; i++) { h[i] = scans_vek[i]->handle(); poses[3 * i] = cfear_tx(T_vek[i]); poses[3 * i + 1] = cfear_ty(T_vek[i]); poses[3 * i + 2] = cfear_yaw(T_vek[i]); }
    ScopedParams sp(dev_, radar_reg->my_params(dev_));  // the samples are GetCost calls of radar_reg (:305)
    double cov[36]; int ok = 0;
    dev_->check(cfear_cov_by_sampling(dev_->ctx(), h.data(), (int)n, poses.data(), (int)radar_reg->itr_, par.cov_sampling_xy_range, par.cov_sampling_yaw_range,
                                      (int)par.cov_sampling_samples_per_axis, par.cov_sampling_covariance_scaler, radar_reg->summary_.final_cost,
                                      radar_reg->summary_.num_residuals, cov, &ok, nullptr), "cfear_cov_by_sampling");
    if (!ok) return false;
    for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) cov_sampled(a, b) = cov[6 * a + b];
    return true;
  }
  DevicePtr dev_; Parameters par;
  Affine3d Tcurrent = cfear_from_xyt(0, 0, 0), T_prev = cfear_from_xyt(0, 0, 0), Tmot = cfear_from_xyt(0, 0, 0);  // odometrykeyframefuser.cpp:34-38
  Matrix6d cov_current = cfear_mat6_identity(); std::vector<Keyframe> keyframes_; size_t nr_callbacks_ = 0;
};

}  // namespace CFEAR_Radarodometry
