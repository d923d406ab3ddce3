// offline_odometry.cpp -- ROS-free counterpart of the reference's src/offline_odometry.cpp main loop
// (:73-127): replays raw polar sweeps at maximum rate through radarDriver::CallbackOffline and
// OdometryKeyframeFuser::pointcloudCallback (both running on the MI355X through the C ABI) and
// writes the estimated trajectory in KITTI format (one 3x4 row-major matrix per line, fixed 6
// decimals, as EvalTrajectory / types.cpp:64-73 do).
//
// Input: a flat binary file of uint8 sweeps, frames x azimuths x range-bins, rows = azimuth
// (radar_driver.cpp:92-98). Flags follow offline_odometry.cpp:152-193 where they exist there.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "cfear_hip/cfear_host.hpp"

using namespace CFEAR_Radarodometry;

static const char* arg(int argc, char** argv, const char* name, const char* def) {
  for (int i = 1; i + 1 < argc; i++) if (!strcmp(argv[i], name)) return argv[i + 1];
  return def;
}

int main(int argc, char** argv) {
  if (argc < 2 || !strcmp(argv[1], "--help") || !strcmp(argv[1], "-h")) {
    printf("usage: offline_odometry --frames sweeps.u8 [--azimuths 400] [--bins 3360] [--range-res 0.0438] [--res 3.5]\n"
           "       [--min_distance 2.5] [--submap_scan_size 3] [--weight_intensity 1] [--k_strongest 12] [--z-min 65]\n"
           "       [--radar_ccw 0] [--disable_compensate 0] [--cost_type P2L] [--loss_type Huber] [--loss_limit 0.1]\n"
           "       [--covar_scale 1] [--regularization 1] [--weight_option 0] [--registered_min_keyframe_dist 1.5]\n"
           "       [--est_directory .] [--device 0] [--filter-type kstrong|CA-CFAR] [--nn-tie 0|1|2] [--voxel-order 0|1]\n"
           "       [--dataset oxford|mulran|...]   anything but oxford: the sweeps in the file are range-major (rows = range bins, bins x azimuths,\n"
           "                      radar_driver.cpp:74-90) and go through the driver's cv::rotate first; --radar_ccw usually goes with it\n"
           "       [--soft_constraint 0] [--covar_sampling 0] [--covar_XY_sample_range 0.4] [--covar_yaw_sample_range 0.0043625]\n"
           "       [--covar_samples_per_axis 3] [--covar_sampling_scale 4]   (offline_odometry.cpp:166-178; --cov_file <path> writes the 36 values per sweep)\n"
           "       [--pinned_frames 1]  the per-sweep route reads every sweep into ONE page-locked buffer (cfear_host_alloc) instead of pageable memory:\n"
           "                      what a reader that owns its frame buffer can do (the image then goes to the device by DMA straight from it); a rosbag\n"
           "                      message is a fresh pageable allocation per sweep - the default 0 measures that\n"
           "       [--replay 1]   whole recording through cfear_odometry_replay_host (pieces of 256 sweeps in pinned memory, no\n"
           "                      host round trip per sweep) instead of one CallbackOffline + pointcloudCallback per sweep\n");
    return argc < 2;
  }
  const std::string frames = arg(argc, argv, "--frames", "");
  const int A = atoi(arg(argc, argv, "--azimuths", "400")), R = atoi(arg(argc, argv, "--bins", "3360"));
  // defaults of offline_odometry.cpp:155-187
  radarDriver::Parameters rad_par;
  rad_par.range_res = (float)atof(arg(argc, argv, "--range-res", "0.0438"));
  rad_par.min_distance = (float)atof(arg(argc, argv, "--min_distance", "2.5"));
  rad_par.k_strongest = atoi(arg(argc, argv, "--k_strongest", "12"));
  rad_par.z_min = (float)atof(arg(argc, argv, "--z-min", "65"));
  rad_par.azimuths = A;
  rad_par.dataset = arg(argc, argv, "--dataset", "oxford");  // offline_odometry.cpp:186, :251
  rad_par.filter_type_ = Str2filter(arg(argc, argv, "--filter-type", "kstrong"));  // offline_odometry.cpp:269
  // the reference's option reuse for the CA-CFAR sweeps (offline_odometry.cpp:260-265): nb_guard_cells <- k_strongest,
  // false_alarm_rate <- regularization, window_size <- covar_scale
  rad_par.nb_guard_cells = rad_par.k_strongest;
  rad_par.false_alarm_rate = (float)atof(arg(argc, argv, "--regularization", "1"));
  rad_par.window_size = (int)atof(arg(argc, argv, "--covar_scale", "1"));
  OdometryKeyframeFuser::Parameters par;
  par.res = atof(arg(argc, argv, "--res", "3.5"));
  par.submap_scan_size = atoi(arg(argc, argv, "--submap_scan_size", "3"));
  par.weight_intensity_ = atoi(arg(argc, argv, "--weight_intensity", "1")) != 0;
  par.radar_ccw = atoi(arg(argc, argv, "--radar_ccw", "0")) != 0;
  par.compensate = atoi(arg(argc, argv, "--disable_compensate", "0")) == 0;
  par.cost_type = arg(argc, argv, "--cost_type", "P2L");
  par.loss_type_ = arg(argc, argv, "--loss_type", "Huber");
  par.loss_limit_ = atof(arg(argc, argv, "--loss_limit", "0.1"));
  par.covar_scale_ = atof(arg(argc, argv, "--covar_scale", "1"));
  par.regularization_ = atof(arg(argc, argv, "--regularization", "1"));
  par.weight_opt = static_cast<weightoption>(atoi(arg(argc, argv, "--weight_option", "0")));
  par.min_keyframe_dist_ = atof(arg(argc, argv, "--registered_min_keyframe_dist", "1.5"));
  par.use_guess = true;  // forced at offline_odometry.cpp:273
  par.soft_constraint = atoi(arg(argc, argv, "--soft_constraint", "0")) != 0;                  // :166, :274
  par.estimate_cov_by_sampling = atoi(arg(argc, argv, "--covar_sampling", "0")) != 0;          // :173, :216-217
  par.cov_sampling_xy_range = atof(arg(argc, argv, "--covar_XY_sample_range", "0.4"));         // :176
  par.cov_sampling_yaw_range = atof(arg(argc, argv, "--covar_yaw_sample_range", "0.0043625")); // :177
  par.cov_sampling_samples_per_axis = (unsigned)atoi(arg(argc, argv, "--covar_samples_per_axis", "3"));  // :178
  par.cov_sampling_covariance_scaler = atof(arg(argc, argv, "--covar_sampling_scale", "4"));   // :179
  const std::string cov_file = arg(argc, argv, "--cov_file", "");
  const std::string est_dir = arg(argc, argv, "--est_directory", ".");

  std::ifstream in(frames, std::ios::binary);
  if (!in) { std::cerr << "cannot open " << frames << std::endl; return 2; }
  cfear_params p; cfear_default_params(&p);
  p.submap_scan_size = par.submap_scan_size;
  try {
    DevicePtr dev(new Device(p, A, R, atoi(arg(argc, argv, "--device", "0"))));
    Device::SetDefault(dev);  // the reference-signature constructors and Compensate(cloud, ...) find this context
    // parity modes (include/cfear_hip.h cfear_tune): --nn-tie 2 = FLANN's kd-tree order among equidistant cells, --voxel-order 1 = PCL <= 1.9's
    // std::sort order inside a voxel - together what an Ubuntu 18.04 build of the reference does where its sources leave the choice to a library
    dev->check(cfear_tune(dev->ctx(), CFEAR_TUNE_NN_TIE_RULE, atoi(arg(argc, argv, "--nn-tie", "0"))), "cfear_tune");
    if (!atoi(arg(argc, argv, "--replay", "0"))) dev->check(cfear_tune(dev->ctx(), CFEAR_TUNE_VOXEL_ORDER, atoi(arg(argc, argv, "--voxel-order", "0"))), "cfear_tune");
    if (atoi(arg(argc, argv, "--replay", "0"))) {
      if (rad_par.dataset != "oxford" || par.soft_constraint || par.estimate_cov_by_sampling)
        throw std::runtime_error("--replay 1 runs the Oxford route without soft constraints / sampled covariances; use the per-sweep route for those");
      // maximum-rate replay: the same parameters the two classes would apply, set once; the loop of offline_odometry.cpp:103-125
      // runs on the device sweep after sweep, the poses of a piece come back together
      cfear_params q = p;
      q.z_min = rad_par.z_min; q.range_res = rad_par.range_res; q.min_distance = rad_par.min_distance; q.k_strongest = rad_par.k_strongest;
      q.res = par.res; q.weight_intensity = par.weight_intensity_ ? 1 : 0; q.submap_scan_size = par.submap_scan_size;
      const cost_metric cm = Str2Cost(par.cost_type);
      q.cost = cm == P2L ? CFEAR_COST_P2L : (cm == P2D ? CFEAR_COST_P2D : CFEAR_COST_P2P);
      q.loss = (int)Str2loss(par.loss_type_); q.loss_limit = par.loss_limit_; q.weight_opt = (int)par.weight_opt;
      q.covar_scale = par.covar_scale_; q.regularization = par.regularization_;
      q.compensate = par.compensate ? 1 : 0; q.radar_ccw = par.radar_ccw ? 1 : 0; q.min_keyframe_dist = par.min_keyframe_dist_;
      if (rad_par.filter_type_ == filtertype::CACFAR) {  // radar_driver.cpp:52-56 in front of the device fuser
        q.filter_type = CFEAR_FILTER_CACFAR; q.cfar_window_size = rad_par.window_size; q.cfar_nb_guard_cells = rad_par.nb_guard_cells;
        q.cfar_false_alarm_rate = rad_par.false_alarm_rate; q.cfar_max_distance = 400.0;
        q.cfar_max_points = atoi(arg(argc, argv, "--cfar_max_points", "0"));
      }
      dev->set_params(q);
      cfear_odometry* odo = nullptr;
      dev->check(cfear_odometry_create(dev->ctx(), 1, &odo), "cfear_odometry_create");
      const int piece = 256;
      const size_t sweep = (size_t)A * R;
      void* pinned = nullptr;
      dev->check(cfear_host_alloc(dev->ctx(), sweep * piece, &pinned), "cfear_host_alloc");
      std::vector<cfear_sweep_record> rec(piece);
      std::ofstream est(est_dir + "/est_00.txt");
      est << std::fixed; est.precision(6);
      int n = 0;
      double t_dev = 0;
      const auto t_start = std::chrono::steady_clock::now();
      for (;;) {
        in.read(static_cast<char*>(pinned), (std::streamsize)(sweep * piece));
        const int got = (int)((size_t)in.gcount() / sweep);
        if (got <= 0) break;
        const auto t0 = std::chrono::steady_clock::now();
        dev->check(cfear_odometry_replay_host(dev->ctx(), odo, static_cast<const uint8_t*>(pinned), got, rec.data()), "cfear_odometry_replay_host");
        t_dev += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        for (int i = 0; i < got; i++) {
          const double c = std::cos(rec[i].pose[2]), sn = std::sin(rec[i].pose[2]);
          est << c << " " << -sn << " 0.000000 " << rec[i].pose[0] << " " << sn << " " << c << " 0.000000 " << rec[i].pose[1] << " "
              << "0.000000 0.000000 1.000000 0.000000\n";
        }
        n += got;
        const double tot = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        std::cout << "Frame: " << n << ", avg: " << n / tot << " Hz (device part alone: " << n / t_dev << " Hz)" << std::endl;  // :125
      }
      std::cout << "frames " << n << std::endl;
      cfear_host_free(dev->ctx(), pinned);
      cfear_odometry_destroy(dev->ctx(), odo);
      return 0;
    }
    radarDriver driver(dev, rad_par, true);
    OdometryKeyframeFuser fuser(dev, par, true);
    const bool pinned_frames = atoi(arg(argc, argv, "--pinned_frames", "0")) != 0;
    std::vector<uint8_t> img_pageable(pinned_frames ? 0 : (size_t)A * R);
    uint8_t* img_pinned = nullptr;
    if (pinned_frames) dev->check(cfear_host_alloc(dev->ctx(), (size_t)A * R, reinterpret_cast<void**>(&img_pinned)), "cfear_host_alloc");
    struct ImgView { uint8_t* p; size_t n; uint8_t* data() const { return p; } size_t size() const { return n; } } img{pinned_frames ? img_pinned : img_pageable.data(), (size_t)A * R};
    std::ofstream est(est_dir + "/est_00.txt");
    est << std::fixed; est.precision(6);
    std::ofstream covs;
    if (!cov_file.empty()) { covs.open(cov_file); covs.precision(17); }
    int n = 0;
    const bool range_major = rad_par.dataset != "oxford";
    double tot = 0;  // the reference's `tot`: callback time only, the bag read is outside (offline_odometry.cpp:99, :119-125)
    const auto t_start = std::chrono::steady_clock::now();
    while (in.read(reinterpret_cast<char*>(img.data()), (std::streamsize)img.size())) {
      const auto t0 = std::chrono::steady_clock::now();
      // Oxford: rows = azimuth (radar_driver.cpp:92-98). The other datasets deliver rows = range bins, and the driver rotates (:84)
      PolarImage pi; pi.rows = range_major ? R : A; pi.cols = range_major ? A : R; pi.data = img.data(); pi.stamp = (uint64_t)n;
      CloudPtr cloud, cloud_peaks;
      driver.CallbackOffline(pi, cloud, cloud_peaks);                     // offline_odometry.cpp:103
      const auto t1 = std::chrono::steady_clock::now();
      CFEAR_TIMING.Document("Filtering", std::chrono::duration<double, std::milli>(t1 - t0).count());  // radar_driver.cpp:111
      CFEAR_TIMING.Document("Filtered points", (double)cloud->size());    // :104
      Affine3d Tcurrent = cfear_from_xyt(0, 0, 0);
      Matrix6d cov_current = cfear_mat6_identity();
      fuser.pointcloudCallback(cloud, cloud_peaks, Tcurrent, pi.stamp, cov_current);   // :108
      if (covs.is_open()) { for (int a = 0; a < 6; a++) for (int b = 0; b < 6; b++) covs << cov_current(a, b) << (a == 5 && b == 5 ? "\n" : " "); }
      const auto t2 = std::chrono::steady_clock::now();
      CFEAR_TIMING.Document("Registration", std::chrono::duration<double, std::milli>(t2 - t1).count());  // odometrykeyframefuser.cpp:404
      double Rm[2][2]; cfear_linear2(Tcurrent, Rm);
      est << Rm[0][0] << " " << Rm[0][1] << " 0.000000 " << cfear_tx(Tcurrent) << " "
          << Rm[1][0] << " " << Rm[1][1] << " 0.000000 " << cfear_ty(Tcurrent) << " "
          << "0.000000 0.000000 1.000000 0.000000\n";
      n++;
      tot += std::chrono::duration<double>(t2 - t0).count();
      if (n % 10 == 0) std::cout << "Frame: " << n << ", dur: " << std::chrono::duration<double>(t2 - t0).count() << ", avg: " << n / tot << " Hz" << std::endl;  // :125
    }
    std::cout << "frames " << n << ", with the file read: " << n / std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() << " Hz\n"
              << CFEAR_TIMING.GetStatistics();
    if (img_pinned) { cfear_synchronize(dev->ctx()); cfear_host_free(dev->ctx(), img_pinned); }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 3;
  }
  return 0;
}
