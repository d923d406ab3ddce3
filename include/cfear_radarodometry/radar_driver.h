// radar_driver.h -- drop-in for the reference's include/cfear_radarodometry/radar_driver.h: put this repository's include/ directory in
// front of the reference's on the include path and link libcfear_hip.so. The classes and functions of the
// hot path that the reference declares in this header come from cfear_host.hpp with the reference's signatures over the
// real ROS / PCL / Eigen / OpenCV types (cfear_types_ros.h); what they replace, line by line, is listed there and in
// include/cfear_hip.h. NOT compiled in this repository's image (no ROS / PCL / Eigen / OpenCV there).
#pragma once
#include "cfear_radarodometry/cfear_types_ros.h"
#include "cfear_hip/cfear_host.hpp"  // (this repository's include/ directory is on the include path: installed as include/cfear_hip/)
// radar_driver.h:24-30,32-120: filtertype, Filter2str, Str2filter, class radarDriver with
//   radarDriver(const Parameters& pars, bool disable_callback = false);                                            (:86)
//   void CallbackOffline(const sensor_msgs::ImageConstPtr&, pcl::PointCloud<pcl::PointXYZI>::Ptr& cloud, ...Ptr& cloud_peaks);  (:90)
//   cv_bridge::CvImagePtr cv_polar_image;                                                                          (:92)
// The ROS subscriptions of the constructor (radar_driver.cpp:31-36, live mode) are not part of the offline path.
