// cfear_types_pod.hpp -- POD stand-ins for the ROS / PCL / Eigen / OpenCV types that cross the reference's interfaces on the
// hot path, and the small adapter functions cfear_host.hpp is written against. This image has none of those libraries; in
// a tree that has them include/cfear_radarodometry/cfear_types_ros.h defines the same names over the real types and the
// same class code (cfear_host.hpp) compiles against them.
#pragma once
#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

#define CFEAR_HOST_TYPES_DEFINED 1
#define CFEAR_SHARED_PTR std::shared_ptr

namespace CFEAR_Radarodometry {

struct PointXYZI { float x = 0, y = 0, z = 0, intensity = 0; };                   // pcl::PointXYZI (32 bytes, 16-aligned there)
struct PointCloudXYZI { std::vector<PointXYZI> points; uint64_t stamp = 0;         // pcl::PointCloud<pcl::PointXYZI>
  size_t size() const { return points.size(); } };
typedef std::shared_ptr<PointCloudXYZI> CloudPtr;                                  // ...::Ptr
// sensor_msgs::Image 8UC1, rows = azimuth (the reference passes sensor_msgs::ImageConstPtr: a pointer-like handle)
struct PolarImage { int rows = 0, cols = 0; const uint8_t* data = nullptr; uint64_t stamp = 0; };
typedef PolarImage ImageConstPtr;
struct CvImage { int rows = 0, cols = 0; std::vector<uint8_t> image; uint64_t stamp = 0; };  // cv_bridge::CvImage
typedef std::shared_ptr<CvImage> CvImagePtr;

struct Vector2d { double x = 0, y = 0; Vector2d() {} Vector2d(double x_, double y_) : x(x_), y(y_) {} double operator()(int i) const { return i ? y : x; } double& operator()(int i) { return i ? y : x; } };
struct Matrix2d { double m[2][2] = {{0, 0}, {0, 0}}; double operator()(int r, int c) const { return m[r][c]; } double& operator()(int r, int c) { return m[r][c]; } };
typedef struct Matrix6dT {
  double m[6][6];
  Matrix6dT() { for (auto& r : m) for (double& v : r) v = 0; for (int i = 0; i < 6; i++) m[i][i] = 1; }
  double operator()(int r, int c) const { return m[r][c]; }
  double& operator()(int r, int c) { return m[r][c]; }
} Matrix6d;

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

// ---- adapters (the only functions of cfear_host.hpp that know what the types look like) ------------------------------------
inline CloudPtr cfear_make_cloud() { return CloudPtr(new PointCloudXYZI()); }
inline size_t cfear_cloud_size(const PointCloudXYZI& c) { return c.points.size(); }
inline void cfear_cloud_to_xyi(const PointCloudXYZI& c, std::vector<float>& xyi) {
  xyi.resize(3 * c.points.size() + 3);
  for (size_t i = 0; i < c.points.size(); i++) { xyi[3 * i] = c.points[i].x; xyi[3 * i + 1] = c.points[i].y; xyi[3 * i + 2] = c.points[i].intensity; }
}
inline void cfear_cloud_from_xyi(PointCloudXYZI& c, const float* xyi, size_t n) {
  c.points.resize(n);
  for (size_t i = 0; i < n; i++) { c.points[i].x = xyi[3 * i]; c.points[i].y = xyi[3 * i + 1]; c.points[i].z = 0; c.points[i].intensity = xyi[3 * i + 2]; }
}
inline bool cfear_image_null(const ImageConstPtr& m) { return m.data == nullptr; }
// the message's own bytes when they already are what toCvCopy(mono8) would produce (8-bit, one channel, rows back to back); else null
inline const uint8_t* cfear_image_raw(const ImageConstPtr& m, int* rows, int* cols) { *rows = m.rows; *cols = m.cols; return m.data; }
inline CvImagePtr cfear_image_to_cv(const ImageConstPtr& m) {  // cv_bridge::toCvCopy(msg, "mono8")
  CvImagePtr c(new CvImage()); c->rows = m.rows; c->cols = m.cols; c->stamp = m.stamp;
  c->image.assign(m.data, m.data + (size_t)m.rows * m.cols);
  return c;
}
inline CvImagePtr cfear_cv_from_buffer(int rows, int cols, std::vector<uint8_t>&& buf, const CvImagePtr& like) {
  CvImagePtr c(new CvImage()); c->rows = rows; c->cols = cols; c->stamp = like ? like->stamp : 0; c->image = std::move(buf);
  return c;
}
inline int cfear_cv_rows(const CvImagePtr& c) { return c->rows; }
inline int cfear_cv_cols(const CvImagePtr& c) { return c->cols; }
inline const uint8_t* cfear_cv_data(const CvImagePtr& c) { return c->image.data(); }
inline void cfear_cloud_stamp_from_cv(PointCloudXYZI& c, const CvImagePtr& img) { c.stamp = img->stamp; }  // pcl_conversions::toPCL(header.stamp, ...)

inline double cfear_tx(const Affine3d& T) { return T.t[0]; }
inline double cfear_ty(const Affine3d& T) { return T.t[1]; }
inline double cfear_yaw(const Affine3d& T) { return T.yaw(); }
inline double cfear_tnorm(const Affine3d& T) { return T.translation_norm(); }
inline Affine3d cfear_from_xyt(double x, double y, double th) { return Affine3d::FromXYT(x, y, th); }
inline void cfear_linear2(const Affine3d& T, double R[2][2]) { for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) R[i][j] = T.l[i][j]; }
inline Vector2d cfear_vec2(double x, double y) { Vector2d v; v.x = x; v.y = y; return v; }
inline Matrix2d cfear_mat2(double a, double b, double c, double d) { Matrix2d m; m.m[0][0] = a; m.m[0][1] = b; m.m[1][0] = c; m.m[1][1] = d; return m; }
inline Matrix6d cfear_mat6_identity() { return Matrix6d(); }

}  // namespace CFEAR_Radarodometry
