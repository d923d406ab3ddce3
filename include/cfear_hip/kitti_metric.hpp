// kitti_metric.hpp -- KITTI odometry drift metric (translation %, rotation deg/100 m) over KITTI-format
// trajectory files (one 3x4 row-major pose per line), the format EvalTrajectory::Write / MatToString produce
// (eval_trajectory.cpp:169-184, types.cpp:64-73). The metric itself is external to the reference (it is what the
// published numbers in BASELINE.md are quoted in): segments of 100..800 m, a new segment start every 10 frames,
// error = delta_est^-1 * delta_gt per segment, averaged over all segments (KITTI devkit evaluate_odometry).
#pragma once
#include <array>
#include <cmath>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

namespace cfear_host {

struct Pose34 { double m[3][4]; };

inline Pose34 pose_mul(const Pose34& a, const Pose34& b) {
  Pose34 o;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) o.m[r][c] = a.m[r][0] * b.m[0][c] + a.m[r][1] * b.m[1][c] + a.m[r][2] * b.m[2][c];
    o.m[r][3] = a.m[r][0] * b.m[0][3] + a.m[r][1] * b.m[1][3] + a.m[r][2] * b.m[2][3] + a.m[r][3];
  }
  return o;
}
inline Pose34 pose_inv(const Pose34& a) {  // general inverse of [A t; 0 1] (6-decimal text poses are not exactly orthonormal)
  const double (*m)[4] = a.m;
  const double c00 = m[1][1] * m[2][2] - m[1][2] * m[2][1], c01 = m[1][2] * m[2][0] - m[1][0] * m[2][2], c02 = m[1][0] * m[2][1] - m[1][1] * m[2][0];
  const double det = m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02, id = 1.0 / det;
  Pose34 o;
  o.m[0][0] = c00 * id; o.m[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) * id; o.m[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) * id;
  o.m[1][0] = c01 * id; o.m[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) * id; o.m[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) * id;
  o.m[2][0] = c02 * id; o.m[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) * id; o.m[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) * id;
  for (int r = 0; r < 3; r++) o.m[r][3] = -(o.m[r][0] * m[0][3] + o.m[r][1] * m[1][3] + o.m[r][2] * m[2][3]);
  return o;
}

inline bool read_kitti_poses(const std::string& path, std::vector<Pose34>& out) {
  std::ifstream in(path);
  if (!in) return false;
  std::string line;
  while (std::getline(in, line)) {
    std::istringstream ss(line);
    Pose34 p;
    bool ok = true;
    for (int r = 0; r < 3 && ok; r++)
      for (int c = 0; c < 4 && ok; c++) ok = static_cast<bool>(ss >> p.m[r][c]);
    if (ok) out.push_back(p);
  }
  return true;
}

struct KittiDrift {
  double translation_percent = 0;     // average translational error, % of the segment length
  double rotation_deg_per_100m = 0;   // average rotational error
  int segments = 0;                   // number of (start, length) pairs that fit into the trajectory
};

inline KittiDrift kitti_drift(const std::vector<Pose34>& gt, const std::vector<Pose34>& est) {
  KittiDrift d;
  const size_t n = gt.size() < est.size() ? gt.size() : est.size();
  if (n < 2) return d;
  std::vector<double> dist(n, 0.0);
  for (size_t i = 1; i < n; i++) {
    const double dx = gt[i].m[0][3] - gt[i - 1].m[0][3], dy = gt[i].m[1][3] - gt[i - 1].m[1][3], dz = gt[i].m[2][3] - gt[i - 1].m[2][3];
    dist[i] = dist[i - 1] + std::sqrt(dx * dx + dy * dy + dz * dz);
  }
  const double lengths[8] = {100, 200, 300, 400, 500, 600, 700, 800};
  const size_t step = 10;
  double sum_t = 0, sum_r = 0;
  for (size_t first = 0; first < n; first += step) {
    for (double len : lengths) {
      size_t last = n;
      for (size_t i = first; i < n; i++)
        if (dist[i] > dist[first] + len) { last = i; break; }
      if (last == n) continue;
      const Pose34 dgt = pose_mul(pose_inv(gt[first]), gt[last]);
      const Pose34 des = pose_mul(pose_inv(est[first]), est[last]);
      const Pose34 e = pose_mul(pose_inv(des), dgt);
      double c = 0.5 * (e.m[0][0] + e.m[1][1] + e.m[2][2] - 1.0);
      c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
      const double r_err = std::acos(c);
      const double t_err = std::sqrt(e.m[0][3] * e.m[0][3] + e.m[1][3] * e.m[1][3] + e.m[2][3] * e.m[2][3]);
      sum_r += r_err / len;
      sum_t += t_err / len;
      d.segments++;
    }
  }
  if (d.segments > 0) {
    d.translation_percent = 100.0 * sum_t / d.segments;
    d.rotation_deg_per_100m = (sum_r / d.segments) * (180.0 / 3.14159265358979323846) * 100.0;
  }
  return d;
}

}  // namespace cfear_host
