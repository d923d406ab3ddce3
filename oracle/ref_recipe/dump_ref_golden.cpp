// dump_ref_golden.cpp -- runs the committed inputs through the REFERENCE's own classes (compiled unmodified from $REF with the
// real ROS / PCL / Ceres / Eigen / OpenCV, see CMakeLists.txt) and writes what the oracle and the HIP path are compared with.
//   dump_ref_golden <ref_inputs.bin> <ref_outputs.bin>          (needs a running roscore: the classes hold ros::NodeHandle members)
// Not compiled in this repository's image. Container format: refio.py.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include <ceres/ceres.h>
#include <ceres/version.h>

#include <ros/ros.h>
#include <cv_bridge/cv_bridge.h>
#include <sensor_msgs/image_encodings.h>

#include "cfear_radarodometry/odometrykeyframefuser.h"
#include "cfear_radarodometry/radar_driver.h"
#include "cfear_radarodometry/radar_filters.h"
#include "cfear_radarodometry/cfar.h"
#include "cfear_radarodometry/n_scan_normal.h"
#include "cfear_radarodometry/pointnormal.h"

using namespace CFEAR_Radarodometry;

struct Arr { int code = 0; std::vector<uint32_t> dims; std::vector<uint8_t> data; };
static const int ITEM[5] = {1, 4, 4, 8, 4};
static std::map<std::string, Arr> read_all(const char* path) {
  std::ifstream in(path, std::ios::binary);
  std::map<std::string, Arr> out;
  uint32_t n;
  while (in.read(reinterpret_cast<char*>(&n), 4)) {
    std::string name(n, ' '); in.read(&name[0], n);
    uint8_t code, nd; in.read(reinterpret_cast<char*>(&code), 1); in.read(reinterpret_cast<char*>(&nd), 1);
    Arr a; a.code = code; a.dims.resize(nd);
    size_t cnt = 1;
    for (int i = 0; i < nd; i++) { in.read(reinterpret_cast<char*>(&a.dims[i]), 4); cnt *= a.dims[i]; }
    a.data.resize(cnt * ITEM[code]); in.read(reinterpret_cast<char*>(a.data.data()), (std::streamsize)a.data.size());
    out[name] = a;
  }
  return out;
}
struct Writer {
  std::ofstream out;
  explicit Writer(const char* p) : out(p, std::ios::binary) {}
  void put(const std::string& name, int code, const std::vector<uint32_t>& dims, const void* data) {
    const uint32_t n = (uint32_t)name.size(); const uint8_t c = (uint8_t)code, nd = (uint8_t)dims.size();
    out.write(reinterpret_cast<const char*>(&n), 4); out.write(name.data(), n); out.write(reinterpret_cast<const char*>(&c), 1); out.write(reinterpret_cast<const char*>(&nd), 1);
    size_t cnt = 1;
    for (uint32_t d : dims) { out.write(reinterpret_cast<const char*>(&d), 4); cnt *= d; }
    out.write(reinterpret_cast<const char*>(data), (std::streamsize)(cnt * ITEM[code]));
  }
  void cloud(const std::string& name, const pcl::PointCloud<pcl::PointXYZI>& c) {
    std::vector<float> v(3 * c.size() + 3);
    for (size_t i = 0; i < c.size(); i++) { v[3 * i] = c.points[i].x; v[3 * i + 1] = c.points[i].y; v[3 * i + 2] = c.points[i].intensity; }
    put(name, 2, {(uint32_t)c.size(), 3}, v.data());
  }
  void f64(const std::string& name, const std::vector<double>& v, const std::vector<uint32_t>& dims) { put(name, 3, dims, v.data()); }
  void i32(const std::string& name, const std::vector<int32_t>& v, const std::vector<uint32_t>& dims) { put(name, 1, dims, v.data()); }
};

static cv_bridge::CvImagePtr to_cv(const Arr& a) {  // uint8 [rows][cols], rows = azimuth (radar_driver.cpp:92-98)
  cv_bridge::CvImagePtr p(new cv_bridge::CvImage());
  p->encoding = sensor_msgs::image_encodings::TYPE_8UC1;
  p->image = cv::Mat((int)a.dims[0], (int)a.dims[1], CV_8UC1, const_cast<uint8_t*>(a.data.data())).clone();
  return p;
}

// the fuser keeps radar_reg / keyframes_ protected (odometrykeyframefuser.h:203-208): a derived class may look at them
struct FuserProbe : public OdometryKeyframeFuser {
  FuserProbe(const Parameters& p) : OdometryKeyframeFuser(p, true) {}
  n_scan_normal_reg& reg() { return *radar_reg; }
  size_t keyframes() { return keyframes_.size(); }
  size_t last_cells() { return keyframes_.empty() ? 0 : keyframes_.back().second->GetSize(); }
};

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s ref_inputs.bin ref_outputs.bin\n", argv[0]); return 2; }
  ros::init(argc, argv, "dump_ref_golden");
  std::map<std::string, Arr> in = read_all(argv[1]);
  Writer w(argv[2]);
  // ---- 1. tiles: StructuredKStrongest (radar_filters.cpp:198-337); min_distance = -1 keeps every selected bin in the cloud,
  //         so cloud and peaks cloud carry the full selection in emission order (ascending (intensity, range) per azimuth)
  const int32_t* kz = reinterpret_cast<const int32_t*>(in["tile_kz"].data.data());
  for (auto& kv : in) {
    if (kv.first.compare(0, 5, "tile_") != 0 || kv.first == "tile_kz") continue;
    for (int c = 0; c < 3; c++) {
      const int k = kz[2 * c], z = kz[2 * c + 1];
      StructuredKStrongest filt(to_cv(kv.second), z, k, -1.0, 0.0438);
      pcl::PointCloud<pcl::PointXYZI>::Ptr cloud(new pcl::PointCloud<pcl::PointXYZI>()), peaks(new pcl::PointCloud<pcl::PointXYZI>());
      filt.getPeaksFilteredPointCloud(cloud, false);
      filt.getPeaksFilteredPointCloud(peaks, true);
      const std::string tag = kv.first.substr(5) + "_k" + std::to_string(k) + "_z" + std::to_string(z);
      w.cloud("tilecloud_" + tag, *cloud); w.cloud("tilepeaks_" + tag, *peaks);
    }
  }
  // ---- 2. the eight world sweeps: filter clouds, compensation, cells, and the whole caller loop for P2L and P2D
  const double* wp = reinterpret_cast<const double*>(in["world_params"].data.data());
  const float range_res = (float)wp[0], min_distance = (float)wp[1]; const int k = (int)wp[2]; const float z_min = (float)wp[3];
  std::vector<pcl::PointCloud<pcl::PointXYZI>::Ptr> clouds, peaks;
  for (int t = 0; t < 8; t++) {
    StructuredKStrongest filt(to_cv(in["sweep_" + std::to_string(t)]), (int)z_min, k, min_distance, range_res);  // radar_driver.cpp:58
    pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>()), p(new pcl::PointCloud<pcl::PointXYZI>());
    filt.getPeaksFilteredPointCloud(c, false); filt.getPeaksFilteredPointCloud(p, true);
    w.cloud("world_cloud_" + std::to_string(t), *c); w.cloud("world_peaks_" + std::to_string(t), *p);
    clouds.push_back(c); peaks.push_back(p);
  }
  {  // Compensate + MapPointNormal on sweep 3 (keys of tests/golden/oracle_golden.npz: world3_*)
    const double* m = reinterpret_cast<const double*>(in["comp_motion"].data.data());
    pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>(*clouds[3]));
    std::vector<double> mot = {m[0], m[1], m[2]};
    Compensate(*c, mot, false);  // utils.cpp:96-107
    w.cloud("world3_cloud_comp", *c);
    MapPointNormal map(c, (float)wp[4], Eigen::Vector2d(0, 0), true, false);
    std::vector<cell> cells = map.GetCells();
    std::vector<double> mean, cov, normal, lmin, lmax, scale; std::vector<int32_t> ns;
    for (auto& cl : cells) {
      mean.push_back(cl.u_(0)); mean.push_back(cl.u_(1)); cov.push_back(cl.cov_(0, 0)); cov.push_back(cl.cov_(0, 1)); cov.push_back(cl.cov_(1, 1));
      normal.push_back(cl.snormal_(0)); normal.push_back(cl.snormal_(1)); lmin.push_back(cl.lambda_min); lmax.push_back(cl.lambda_max);
      scale.push_back(cl.scale_); ns.push_back((int32_t)cl.Nsamples_);
    }
    const uint32_t n = (uint32_t)cells.size();
    w.f64("world3_cells_mean", mean, {n, 2}); w.f64("world3_cells_cov", cov, {n, 3}); w.f64("world3_cells_normal", normal, {n, 2});
    w.f64("world3_cells_lambda_min", lmin, {n}); w.f64("world3_cells_lambda_max", lmax, {n}); w.f64("world3_cells_scale", scale, {n});
    w.i32("world3_cells_nsamples", ns, {n});
    // which cell the reference's 1-NN search (KdTreeFLANN<PointXY>::nearestKSearch behind GetClosestIdx, pointnormal.cpp:238-254)
    // returns for a query AT every cell mean: its own index where the float mean is unique, and FLANN's choice among the cells that
    // share a float mean (1-5 % of a scan's cells: DESIGN.md section 2, [3P] sensitivity) - the one thing that pins the tie order
    std::vector<int32_t> self(n, -1);
    for (uint32_t i = 0; i < n; i++) {
      const std::vector<int> r = map.GetClosestIdx(Eigen::Vector2d(cells[i].u_(0), cells[i].u_(1)), 0.5);
      self[i] = r.empty() ? -1 : (int32_t)r[0];
    }
    w.i32("world3_closest_self", self, {n});
  }
  for (int pass = 0; pass < 2; pass++) {  // offline_odometry.cpp:103-108 with the fixture's parameters
    OdometryKeyframeFuser::Parameters par;
    par.cost_type = pass == 0 ? "P2L" : "P2D"; par.loss_type_ = "Huber"; par.loss_limit_ = wp[6];
    par.weight_opt = weightoption::Combined_weights; par.submap_scan_size = (int)wp[5]; par.res = wp[4]; par.weight_intensity_ = true;
    par.compensate = true; par.radar_ccw = false; par.use_guess = true; par.min_keyframe_dist_ = wp[7];
    par.covar_scale_ = 1.0; par.regularization_ = 0.1;  // cfear_default_params (cabi.hip): the values the oracle fixture was made with
    FuserProbe fuser(par);
    std::vector<double> traj, cost; std::vector<int32_t> outer, last_inner, nres, ncells, nkf;
    for (int t = 0; t < 8; t++) {
      pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>(*clouds[t])), p(new pcl::PointCloud<pcl::PointXYZI>(*peaks[t]));
      Eigen::Affine3d T = Eigen::Affine3d::Identity();
      fuser.pointcloudCallback(c, p, T, ros::Time(1.0 + 0.25 * t));
      std::vector<double> v; Affine3dToVectorXYeZ(T, v);
      traj.insert(traj.end(), v.begin(), v.end());
      outer.push_back((int32_t)fuser.reg().itr_);  // registration.h:107
      last_inner.push_back((int32_t)fuser.reg().summary_.iterations.size());
      cost.push_back(fuser.reg().summary_.final_cost); nres.push_back(fuser.reg().summary_.num_residuals);
      nkf.push_back((int32_t)fuser.keyframes());
    }
    const std::string tag = pass == 0 ? "p2l" : "p2d";
    w.f64("traj_" + tag, traj, {8, 3}); w.i32("outer_" + tag, outer, {8}); w.i32("last_inner_" + tag, last_inner, {8});
    w.f64("final_cost_" + tag, cost, {8}); w.i32("num_residuals_" + tag, nres, {8}); w.i32("keyframes_" + tag, nkf, {8});
  }
  // ---- 3. n_scan_normal_reg called directly (not through the fuser): Register -> poses, itr_, summary_, reg_cov.back()
  //         (GetCovariance, n_scan_normal.cpp:392-433), then GetCost at the registered poses (residual vector + score,
  //         n_scan_normal.cpp:188-213) - for every cost and for the losses the fuser run above does not use
  {
    const double* rp = reinterpret_cast<const double*>(in["reg_poses"].data.data());  // 4 x (x, y, theta): the first guesses
    std::vector<MapNormalPtr> scans;
    for (int t = 0; t < 4; t++) scans.push_back(MapNormalPtr(new MapPointNormal(clouds[t], (float)wp[4], Eigen::Vector2d(0, 0), true, false)));
    struct Cfg { const char* tag; cost_metric cost; loss_type loss; double limit; };
    const Cfg cfgs[] = {{"p2l_huber", P2L, Huber, 0.1}, {"p2l_cauchy", P2L, Cauchy, 0.2}, {"p2l_tukey", P2L, Tukey, 0.5}, {"p2d_huber", P2D, Huber, 0.1},
                        {"p2p_huber", P2P, Huber, 0.1}, {"p2l_softlone", P2L, SoftLOne, 0.1}, {"p2l_none", P2L, None, 0.1}};
    for (const Cfg& c : cfgs) {
      n_scan_normal_reg reg(c.cost, c.loss, c.limit, weightoption::Combined_weights);
      reg.SetD2dPar(1.0, 0.1);  // covar_scale, regularization of the fixture
      std::vector<Eigen::Affine3d> T;
      for (int t = 0; t < 4; t++) T.push_back(vectorToAffine3d(rp[3 * t], rp[3 * t + 1], 0, 0, 0, rp[3 * t + 2]));
      std::vector<Matrix6d> cov(4, Matrix6d::Identity());
      const bool ok = reg.Register(scans, T, cov, false);
      std::vector<double> poses, cv(36), info = {ok ? 1.0 : 0.0, (double)reg.itr_, (double)reg.summary_.iterations.size(), reg.summary_.final_cost,
                                                (double)reg.summary_.num_residuals, reg.getScore()};
      for (int t = 0; t < 4; t++) { std::vector<double> v; Affine3dToVectorXYeZ(T[t], v); poses.insert(poses.end(), v.begin(), v.end()); }
      for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) cv[6 * a + b] = cov.back()(a, b);
      const std::string tag = c.tag;
      w.f64("reg_poses_" + tag, poses, {4, 3}); w.f64("reg_info_" + tag, info, {6}); w.f64("reg_cov_" + tag, cv, {6, 6});
      double score = 0; std::vector<double> residuals;
      const bool cost_ok = reg.GetCost(scans, T, score, residuals);  // itr_ of the object is what Register left (radius 2 m unless 1)
      residuals.push_back(0.0);  // (never an empty record)
      w.f64("getcost_score_" + tag, {cost_ok ? 1.0 : 0.0, score}, {2});
      w.f64("getcost_residuals_" + tag, residuals, {(uint32_t)residuals.size()});
    }
  }
  // ---- 4. AzimuthCACFAR::getFilteredPointCloud (cfar.cpp:27-87) on sweep 0 with the defaults of radarDriver::Parameters
  //         (window_size 10, nb_guard_cells 20, false_alarm_rate 0.01, radar_driver.h:43-44) and max_distance 400
  {
    const double* cp = reinterpret_cast<const double*>(in["cfar_params"].data.data());  // window guard false_alarm max_distance
    AzimuthCACFAR filt((int)cp[0], cp[2], (int)cp[1], range_res, z_min, min_distance, cp[3]);
    pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>());
    filt.getFilteredPointCloud(to_cv(in["sweep_0"]), c);
    w.cloud("cfar_cloud_0", *c);
  }
  // ---- 5. the fuser with the cost-sampling covariance (approximateCovarianceBySampling, odometrykeyframefuser.cpp:261-380, a
  //         private member: reached through par.estimate_cov_by_sampling and the five-argument pointcloudCallback, which hands
  //         back cov_current) and, from a second run without it, the registration covariance of every sweep
  for (int pass = 0; pass < 2; pass++) {
    OdometryKeyframeFuser::Parameters par;
    par.cost_type = "P2L"; par.loss_type_ = "Huber"; par.loss_limit_ = wp[6];
    par.weight_opt = weightoption::Combined_weights; par.submap_scan_size = (int)wp[5]; par.res = wp[4]; par.weight_intensity_ = true;
    par.compensate = true; par.radar_ccw = false; par.use_guess = true; par.min_keyframe_dist_ = wp[7];
    par.covar_scale_ = 1.0; par.regularization_ = 0.1;
    par.estimate_cov_by_sampling = pass == 1; par.cov_samples_to_file_as_well = false;
    FuserProbe fuser(par);
    std::vector<double> covs;
    for (int t = 0; t < 8; t++) {
      pcl::PointCloud<pcl::PointXYZI>::Ptr c(new pcl::PointCloud<pcl::PointXYZI>(*clouds[t])), p(new pcl::PointCloud<pcl::PointXYZI>(*peaks[t]));
      Eigen::Affine3d T = Eigen::Affine3d::Identity();
      Covariance cov = Covariance::Identity();
      fuser.pointcloudCallback(c, p, T, ros::Time(1.0 + 0.25 * t), cov);
      for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) covs.push_back(cov(a, b));
    }
    w.f64(pass == 0 ? "fuser_reg_cov_p2l" : "fuser_sampled_cov_p2l", covs, {8, 6, 6});
  }
  // ---- 6. what the reference leaves to the installed Ceres (CMakeLists.txt:38: find_package(Ceres REQUIRED), no version). Two things differ
  //         between the versions its README's distributions ship (1.13 / 1.14 on Ubuntu 18.04, 2.0 on 20.04) and are visible in the results:
  //         TukeyLoss - rho = a^2/3 (1 - (1 - s/a^2)^3) in 2.0, a^2/6 (...) with rho' = 1/2 (1 - s/a^2)^2 in <= 1.14 (a factor two on every cost and
  //         covariance of a Tukey run; the oracle states the 2.0 form) - and the order of the convergence tests in the trust-region loop
  //         (>= 1.13: the tolerances are checked before the step is accepted; the oracle's order), which the iteration counts above show. The
  //         version and the losses' own Evaluate(s) are written out so that the comparison says which form the box had.
  {
    w.f64("ceres_version", {(double)CERES_VERSION_MAJOR, (double)CERES_VERSION_MINOR, (double)CERES_VERSION_REVISION}, {3});
    const double ss[7] = {0.0, 0.005, 0.01, 0.04, 0.2, 0.25, 1.0};
    ceres::HuberLoss huber(0.1); ceres::CauchyLoss cauchy(0.2); ceres::SoftLOneLoss softl1(0.1); ceres::TukeyLoss tukey(0.5);
    ceres::LossFunction* fs[4] = {&huber, &cauchy, &softl1, &tukey};
    std::vector<double> probe;
    for (int f = 0; f < 4; f++)
      for (int i = 0; i < 7; i++) { double rho[3]; fs[f]->Evaluate(ss[i], rho); probe.insert(probe.end(), rho, rho + 3); }
    w.f64("ceres_loss_probe", probe, {4, 7, 3});  // [Huber 0.1, Cauchy 0.2, SoftLOne 0.1, Tukey 0.5][s][rho, rho', rho'']
  }
  std::printf("wrote %s\n", argv[2]);
  return 0;
}
