// utils.h -- drop-in for the reference's include/cfear_radarodometry/utils.h: put this repository's include/ directory in
// front of the reference's on the include path and link libcfear_hip.so. The classes and functions of the
// hot path that the reference declares in this header come from cfear_host.hpp with the reference's signatures over the
// real ROS / PCL / Eigen / OpenCV types (cfear_types_ros.h); what they replace, line by line, is listed there and in
// include/cfear_hip.h. NOT compiled in this repository's image (no ROS / PCL / Eigen / OpenCV there).
#pragma once
#include "cfear_radarodometry/cfear_types_ros.h"
#include "cfear_hip/cfear_host.hpp"  // (this repository's include/ directory is on the include path: installed as include/cfear_hip/)
// utils.h:47-51: Compensate(cloud, mot | Tmotion, ccw), Affine3dToVectorXYeZ; registration.h: vectorToAffine3d.
