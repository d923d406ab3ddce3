// pointnormal.h -- drop-in for the reference's include/cfear_radarodometry/pointnormal.h: put this repository's include/ directory in
// front of the reference's on the include path and link libcfear_hip.so. The classes and functions of the
// hot path that the reference declares in this header come from cfear_host.hpp with the reference's signatures over the
// real ROS / PCL / Eigen / OpenCV types (cfear_types_ros.h); what they replace, line by line, is listed there and in
// include/cfear_hip.h. NOT compiled in this repository's image (no ROS / PCL / Eigen / OpenCV there).
#pragma once
#include "cfear_radarodometry/cfear_types_ros.h"
#include "cfear_hip/cfear_host.hpp"  // (this repository's include/ directory is on the include path: installed as include/cfear_hip/)
// pointnormal.h:45-105 class cell, :110-243 class MapPointNormal (both constructors :118,:120; GetCells, GetCell, GetClosest,
// GetClosestIdx, GetCellRelTimeStamp, TransformCells, TransformMap, GetMean2d / GetCov2d / GetNormal2d, GetScan, GetSize,
// static downsample_factor). Boost serialization of cells and maps keeps the reference's archive layout (the .sgh export of types.cpp:103-130
// works on these classes). The RViz publishers (PublishMap ..., pointnormal.cpp:363-512) are not on the path and not provided.
