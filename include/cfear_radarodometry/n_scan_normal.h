// n_scan_normal.h -- drop-in for the reference's include/cfear_radarodometry/n_scan_normal.h: put this repository's include/ directory in
// front of the reference's on the include path and link libcfear_hip.so. The classes and functions of the
// hot path that the reference declares in this header come from cfear_host.hpp with the reference's signatures over the
// real ROS / PCL / Eigen / OpenCV types (cfear_types_ros.h); what they replace, line by line, is listed there and in
// include/cfear_hip.h. NOT compiled in this repository's image (no ROS / PCL / Eigen / OpenCV there).
#pragma once
#include "cfear_radarodometry/cfear_types_ros.h"
#include "cfear_hip/cfear_host.hpp"  // (this repository's include/ directory is on the include path: installed as include/cfear_hip/)
// n_scan_normal.h:27-85 class n_scan_normal_reg: both constructors (:33,:35), Register (:37), GetCost (:41), getScore (:47,:51),
// GetCovarianceScaler (:49), SetD2dPar (:53), SetParameters (:55), public summary_ / itr_ (registration.h:107-110).
// RegisterTimeContinuous and GetSurface (off by default) are not provided.
