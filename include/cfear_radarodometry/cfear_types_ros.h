// cfear_types_ros.h -- the real ROS / PCL / Eigen / OpenCV types behind the drop-in headers of this directory, and the adapter
// functions cfear_host.hpp is written against (same names as cfear_radarodometry_code_public_amd/host/cfear_types_pod.hpp,
// which is what this repository's own tests compile: its image has none of these libraries, so THIS FILE HAS NOT BEEN
// COMPILED HERE). Needs roscpp, sensor_msgs, cv_bridge, pcl_ros / pcl_conversions, Eigen3, OpenCV, boost.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include <boost/shared_ptr.hpp>
#include <Eigen/Eigen>
#include <cv_bridge/cv_bridge.h>
#include <opencv2/core.hpp>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl_conversions/pcl_conversions.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/image_encodings.h>

#include "cfear_radarodometry/statistics.h"  // the reference's own header and statistics.cpp stay in the build (not on the hot path)

#define CFEAR_HOST_TYPES_DEFINED 1
#define CFEAR_SHARED_PTR boost::shared_ptr
#define CFEAR_TIMING CFEAR_Radarodometry::timing  // statistics.h: the reference's global timing object
// boost archives of cells / maps (pointnormal.h:86-101, :201-226) keep the reference's layout, so SaveSimpleGraph / LoadSimpleGraph
// (types.cpp:103-130: the .sgh export) work unchanged on the drop-in classes; the Eigen / PCL adaptors are the reference's own
#include <boost/serialization/serialization.hpp>
#include <boost/serialization/split_member.hpp>
#include <boost/serialization/vector.hpp>
#include "cfear_radarodometry/serialization.h"
#define CFEAR_HOST_BOOST_SERIALIZATION 1
#define CFEAR_DOWNSAMPLED_PTR pcl::PointCloud<pcl::PointXY>::Ptr

namespace CFEAR_Radarodometry {

typedef pcl::PointXYZI PointXYZI;
typedef pcl::PointCloud<pcl::PointXYZI> PointCloudXYZI;
typedef pcl::PointCloud<pcl::PointXYZI>::Ptr CloudPtr;
typedef sensor_msgs::ImageConstPtr ImageConstPtr;
typedef cv_bridge::CvImagePtr CvImagePtr;
typedef Eigen::Vector2d Vector2d;
typedef Eigen::Matrix2d Matrix2d;
typedef Eigen::Matrix<double, 6, 6> Matrix6d;  // registration.h:41
typedef Eigen::Affine3d Affine3d;

inline CloudPtr cfear_make_cloud() { return CloudPtr(new PointCloudXYZI()); }
inline size_t cfear_cloud_size(const PointCloudXYZI& c) { return c.points.size(); }
inline void cfear_cloud_to_xyi(const PointCloudXYZI& c, std::vector<float>& xyi) {
  xyi.resize(3 * c.points.size() + 3);
  for (size_t i = 0; i < c.points.size(); i++) { xyi[3 * i] = c.points[i].x; xyi[3 * i + 1] = c.points[i].y; xyi[3 * i + 2] = c.points[i].intensity; }
}
inline void cfear_cloud_from_xyi(PointCloudXYZI& c, const float* xyi, size_t n) {
  c.points.resize(n); c.width = (uint32_t)n; c.height = 1; c.is_dense = true;
  for (size_t i = 0; i < n; i++) { pcl::PointXYZI p; p.x = xyi[3 * i]; p.y = xyi[3 * i + 1]; p.z = 0; p.intensity = xyi[3 * i + 2]; c.points[i] = p; }  // radar_filters.cpp:326-334
}
inline bool cfear_image_null(const ImageConstPtr& m) { return m == NULL; }
// the message's own bytes when they already are what toCvCopy(TYPE_8UC1) would produce (8-bit, one channel, rows back to back); else null
inline const uint8_t* cfear_image_raw(const ImageConstPtr& m, int* rows, int* cols) {
  *rows = (int)m->height; *cols = (int)m->width;
  const bool mono = m->encoding == sensor_msgs::image_encodings::MONO8 || m->encoding == sensor_msgs::image_encodings::TYPE_8UC1;
  return (mono && m->step == m->width && m->data.size() >= (size_t)m->height * m->width) ? m->data.data() : nullptr;
}
// radar_driver.cpp:81-82 / :104-105: toCvCopy(..., MONO8 / TYPE_8UC1); a stamp below 1 ms becomes ros::Time::now()
inline CvImagePtr cfear_image_to_cv(const ImageConstPtr& m) {
  CvImagePtr c = cv_bridge::toCvCopy(m, sensor_msgs::image_encodings::TYPE_8UC1);
  c->header.stamp = m->header.stamp.toSec() < 0.001 ? ros::Time::now() : m->header.stamp;
  if (!c->image.isContinuous()) c->image = c->image.clone();
  return c;
}
inline CvImagePtr cfear_cv_from_buffer(int rows, int cols, std::vector<uint8_t>&& buf, const CvImagePtr& like) {
  CvImagePtr c(new cv_bridge::CvImage());
  if (like) { c->header = like->header; c->encoding = like->encoding; }
  c->image = cv::Mat(rows, cols, CV_8UC1, buf.data()).clone();
  return c;
}
inline int cfear_cv_rows(const CvImagePtr& c) { return c->image.rows; }
inline int cfear_cv_cols(const CvImagePtr& c) { return c->image.cols; }
inline const uint8_t* cfear_cv_data(const CvImagePtr& c) { return c->image.ptr<uint8_t>(0); }
inline void cfear_cloud_stamp_from_cv(PointCloudXYZI& c, const CvImagePtr& img) { pcl_conversions::toPCL(img->header.stamp, c.header.stamp); }  // radar_driver.cpp:66-67

// downsampled_ of the reference (pointnormal.cpp:151-158): the float cell means as a PointXY cloud (template: `cell` is declared later)
template <class CellVector>
inline pcl::PointCloud<pcl::PointXY>::Ptr cfear_make_downsampled(const CellVector& cells) {
  pcl::PointCloud<pcl::PointXY>::Ptr d(new pcl::PointCloud<pcl::PointXY>());
  for (const auto& c : cells) { pcl::PointXY p; p.x = (float)c.u_(0); p.y = (float)c.u_(1); d->push_back(p); }
  return d;
}

inline double cfear_tx(const Affine3d& T) { return T.translation()(0); }
inline double cfear_ty(const Affine3d& T) { return T.translation()(1); }
// Affine3dToVectorXYeZ (utils.cpp:115-122) takes eulerAngles(0,1,2)(2); for the planar poses of this path that is the yaw
inline double cfear_yaw(const Affine3d& T) { return std::atan2(T.linear()(1, 0), T.linear()(1, 1)); }
inline double cfear_tnorm(const Affine3d& T) { return T.translation().norm(); }
inline Affine3d cfear_from_xyt(double x, double y, double th) {  // vectorToAffine3d (registration.cpp:130-136)
  return Eigen::Translation<double, 3>(x, y, 0) * Eigen::AngleAxis<double>(th, Eigen::Vector3d::UnitZ());
}
inline void cfear_linear2(const Affine3d& T, double R[2][2]) { for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) R[i][j] = T.linear()(i, j); }
inline Vector2d cfear_vec2(double x, double y) { return Vector2d(x, y); }
inline Matrix2d cfear_mat2(double a, double b, double c, double d) { Matrix2d m; m << a, b, c, d; return m; }
inline Matrix6d cfear_mat6_identity() { return Matrix6d::Identity(); }

}  // namespace CFEAR_Radarodometry
