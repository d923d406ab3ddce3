// api_check -- exercises the C++ mirror classes of cfear_host.hpp the way reference code uses them (radarDriver with both
// filter types, MapPointNormal and its accessors, n_scan_normal_reg::Register with and without soft constraints, GetCost,
// GetCovarianceScaler) on three sweeps read from a raw uint8 file and prints the numbers as one JSON object;
// tests/test_host_cpp.py compares them with the oracle.   usage: api_check <sweeps.u8> [A R range_res]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>

#include "cfear_hip/cfear_host.hpp"

using namespace CFEAR_Radarodometry;

int main(int argc, char** argv) {
  if (argc < 2) { std::fprintf(stderr, "usage: %s <sweeps.u8> [A R range_res]\n", argv[0]); return 2; }
  const int A = argc > 2 ? atoi(argv[2]) : 400, R = argc > 3 ? atoi(argv[3]) : 3360;
  const float rr = argc > 4 ? (float)atof(argv[4]) : 0.0595238f;
  std::ifstream in(argv[1], std::ios::binary);
  std::vector<std::vector<uint8_t>> imgs;
  for (;;) { std::vector<uint8_t> img((size_t)A * R); if (!in.read(reinterpret_cast<char*>(img.data()), (std::streamsize)img.size())) break; imgs.push_back(img); }
  if (imgs.size() < 3) { std::fprintf(stderr, "need three sweeps\n"); return 2; }
  try {
    // reference-signature constructors throughout (radar_driver.h:86, pointnormal.h:118, n_scan_normal.h:35): the device
    // context is the process-wide default, created when the driver sees its first image
    radarDriver::Parameters rp; rp.range_res = rr; rp.z_min = 60; rp.k_strongest = 12; rp.min_distance = 2.5f;
    radarDriver driver(rp, true);
    std::vector<MapNormalPtr> scans; std::vector<size_t> npts;
    for (int t = 0; t < 3; t++) {
      PolarImage pi; pi.rows = A; pi.cols = R; pi.data = imgs[t].data(); pi.stamp = (uint64_t)t;
      CloudPtr cloud, peaks;
      driver.CallbackOffline(pi, cloud, peaks);
      npts.push_back(cloud->size());
      scans.push_back(MapNormalPtr(new MapPointNormal(cloud, 3.0f, Vector2d(0, 0), true, false)));
    }
    // CA-CFAR through the same driver class (filter_type switch, radar_driver.cpp:52-56)
    radarDriver::Parameters rc = rp; rc.filter_type_ = Str2filter("CA-CFAR");
    radarDriver cfar_driver(rc, true);
    PolarImage pi0; pi0.rows = A; pi0.cols = R; pi0.data = imgs[0].data();
    CloudPtr ccloud, cpeaks;
    cfar_driver.CallbackOffline(pi0, ccloud, cpeaks);
    MapNormalPtr m = scans[2];
    const cell& c0 = m->GetCell(0);
    const std::vector<int> nn = m->GetClosestIdx(c0.u_, 1.0);
    Affine3d Tt = cfear_from_xyt(1.0, -2.0, 0.3);
    std::vector<cell> tc = m->TransformCells(Tt);
    n_scan_normal_reg reg(P2L, Huber, 0.1, Combined_weights);
    // a second registration object with other settings on the same device, and a map built with other settings after
    // the first ones: neither may change what `reg` and `scans` compute (per-object parameter snapshots)
    n_scan_normal_reg other(P2P, None, 0.5, Uniform);
    other.SetD2dPar(3.0, 0.7);
    MapNormalPtr coarse(new MapPointNormal(driver.device_cloud()->dev, *driver.device_cloud(), 5.0f, Vector2d(0, 0), false, false));
    MapNormalPtr raw(new MapPointNormal(driver.device_cloud()->dev, *driver.device_cloud(), 3.0f, Vector2d(0, 0), false, true));  // identity cells
    std::vector<cell> raw_cells = raw->GetCells();
    const std::vector<int> raw_nn = raw->GetClosestIdx(raw_cells[5].u_, 0.5);
    MapNormalPtr moved = m->TransformMap(Tt);  // the transformed-copy constructor (pointnormal.h:120)
    const std::vector<int> moved_nn = moved->GetClosestIdx(tc[0].u_, 0.5);
    std::vector<Affine3d> T = {cfear_from_xyt(0, 0, 0), cfear_from_xyt(1.0, 0.02, 0.02), cfear_from_xyt(2.2, 0.1, 0.05)};
    std::vector<Matrix6d> cov(3);
    std::vector<Affine3d> T1 = T, To = T; std::vector<Matrix6d> covo(3);
    const bool ok_other = other.Register(scans, To, covo, false);
    const bool ok = reg.Register(scans, T1, cov, false);
    double cov_scale = 0; const bool has_scale = reg.GetCovarianceScaler(cov_scale);
    double score = 0; std::vector<double> residuals;
    std::vector<Affine3d> Tq = T1;
    const bool cost_ok = reg.GetCost(scans, Tq, score, residuals);
    const double score_after_get_cost = reg.getScore();
    std::vector<Affine3d> T2 = T; std::vector<Matrix6d> cov2(3);
    for (int a = 0; a < 6; a++) cov2[2](a, a) = 0.05 * 0.05;
    const bool ok_soft = reg.Register(scans, T2, cov2, true);
    // the two calls of odometrykeyframefuser.cpp that used to need an #if 0 (:191 summary_.FullReport(), :210 MapPointNormal::PublishMap)
    const std::string report = reg.summary_.FullReport();
    MapPointNormal::PublishMap("/current_normals", m, T1[2], "sensor_est", -1, 0.5);
    if (report.find("outer iterations") == std::string::npos) throw std::runtime_error("FullReport is empty");
    // Device twins (cfear_host.hpp): a cloud the driver made is compensated / turned into a map on the device without an upload - unless
    // the caller changed it in between, which must be noticed. Every case is compared with the same operation on a copy of the cloud at
    // another address (no twin: the upload route); the results have to be identical bit for bit.
    auto maxdiff = [](const PointCloudXYZI& a, const PointCloudXYZI& b) {
      if (a.points.size() != b.points.size()) return 1e30;
      double m = 0;
      for (size_t i = 0; i < a.points.size(); i++) {
        m = std::fmax(m, std::fabs((double)a.points[i].x - b.points[i].x)); m = std::fmax(m, std::fabs((double)a.points[i].y - b.points[i].y));
        m = std::fmax(m, std::fabs((double)a.points[i].intensity - b.points[i].intensity));
      }
      return m;
    };
    PolarImage pi2; pi2.rows = A; pi2.cols = R; pi2.data = imgs[2].data(); pi2.stamp = 2;
    const Affine3d Tm = cfear_from_xyt(0.8, -0.1, 0.03), Tm2 = cfear_from_xyt(-0.3, 0.2, -0.01);
    CloudPtr ca, pa; driver.CallbackOffline(pi2, ca, pa);
    CloudPtr cb(new PointCloudXYZI(*ca)), pb(new PointCloudXYZI(*pa));
    Compensate(*ca, Tm, false); Compensate(*pa, Tm, false);  // twin route, the sibling compensated ahead
    Compensate(*cb, Tm, false); Compensate(*pb, Tm, false);  // upload route
    const double tw_same = std::fmax(maxdiff(*ca, *cb), maxdiff(*pa, *pb));
    CloudPtr cm, pm; driver.CallbackOffline(pi2, cm, pm);
    cm->points[5].x += 1.0f; cm->points.pop_back();          // the caller edits the cloud between CallbackOffline and Compensate
    CloudPtr cmref(new PointCloudXYZI(*cm));
    Compensate(*cm, Tm, false); Compensate(*cmref, Tm, false);
    const double tw_mutated = maxdiff(*cm, *cmref);
    const bool tw_mutated_differs = cm->points.size() + 1 == ca->points.size() && std::fabs(cm->points[5].x - ca->points[5].x) > 0.5;
    CloudPtr c3, p3; driver.CallbackOffline(pi2, c3, p3);
    Compensate(*c3, Tm, false);                               // compensates p3's twin ahead ...
    p3->points[0].y -= 2.0f;                                  // ... but the caller changes p3 before its own call
    CloudPtr p3ref(new PointCloudXYZI(*p3));
    Compensate(*p3, Tm, false); Compensate(*p3ref, Tm, false);
    const double tw_sibling_mutated = maxdiff(*p3, *p3ref);
    CloudPtr c4, p4; driver.CallbackOffline(pi2, c4, p4);
    CloudPtr p4ref(new PointCloudXYZI(*p4));
    Compensate(*c4, Tm, false); Compensate(*p4, Tm2, true);   // the sibling gets another motion than the one run ahead
    Compensate(*p4ref, Tm2, true);
    const double tw_sibling_other_motion = maxdiff(*p4, *p4ref);
    CloudPtr c5, p5; driver.CallbackOffline(pi2, c5, p5);
    for (size_t i = 0; i < c5->points.size(); i += 2) c5->points[i].x += 0.75f;  // a map of an edited cloud = the map of its copy
    CloudPtr c5ref(new PointCloudXYZI(*c5));
    MapPointNormal m5(c5, 3.0f, Vector2d(0, 0), true, false), m5ref(c5ref, 3.0f, Vector2d(0, 0), true, false);
    MapPointNormal m2twin(ca, 3.0f, Vector2d(0, 0), true, false), m2up(cb, 3.0f, Vector2d(0, 0), true, false);  // compensated: twin vs upload
    double tw_map = std::fabs((double)m5.GetSize() - (double)m5ref.GetSize()) + std::fabs((double)m2twin.GetSize() - (double)m2up.GetSize());
    for (size_t i = 0; i < m5.GetSize() && i < m5ref.GetSize(); i++) tw_map = std::fmax(tw_map, std::fabs(m5.GetCell(i).u_(0) - m5ref.GetCell(i).u_(0)));
    for (size_t i = 0; i < m2twin.GetSize() && i < m2up.GetSize(); i++) tw_map = std::fmax(tw_map, std::fabs(m2twin.GetCell(i).u_(1) - m2up.GetCell(i).u_(1)));
    const bool tw_map_differs = m5.GetSize() != scans[2]->GetSize() || std::fabs(m5.GetCell(0).u_(0) - scans[2]->GetCell(0).u_(0)) > 1e-6;
    std::printf("{\"points\": [%zu, %zu, %zu], \"cfar_points\": %zu, \"cfar_peaks\": %zu, \"cells\": [%zu, %zu, %zu], "
                "\"cell0\": [%.12g, %.12g, %.12g, %.12g], \"closest_self\": %d, \"rel_time0\": %.12g, "
                "\"tcell0\": [%.12g, %.12g, %.12g, %.12g, %.12g], "
                "\"register\": {\"ok\": %d, \"pose\": [%.12g, %.12g, %.12g], \"itr\": %zu, \"score\": %.12g, \"cov00\": %.12g, \"cov_scale\": %.12g, \"has_scale\": %d}, "
                "\"get_cost\": {\"ok\": %d, \"score\": %.12g, \"n\": %zu, \"r0\": %.12g, \"getScore\": %.12g}, "
                "\"register_soft\": {\"ok\": %d, \"pose\": [%.12g, %.12g, %.12g], \"residuals\": %d}, "
                "\"other\": {\"ok\": %d, \"pose\": [%.12g, %.12g, %.12g], \"residuals\": %d}, \"coarse_cells\": %zu, "
                "\"raw\": {\"cells\": %zu, \"points\": %zu, \"nn5\": %d, \"scale\": %.12g, \"cov00\": %.12g}, \"moved\": {\"cells\": %zu, \"nn0\": %d}, "
                "\"twins\": {\"same\": %.9g, \"mutated\": %.9g, \"mutated_differs\": %d, \"sibling_mutated\": %.9g, \"sibling_other_motion\": %.9g, \"map\": %.9g, \"map_differs\": %d, "
                "\"comp0\": [%.9g, %.9g]}}\n",
                npts[0], npts[1], npts[2], ccloud->size(), cpeaks->size(), scans[0]->GetSize(), scans[1]->GetSize(), scans[2]->GetSize(),
                c0.u_(0), c0.u_(1), c0.snormal_(0), c0.snormal_(1), nn.empty() ? -1 : nn[0], m->GetCellRelTimeStamp(0, false),
                tc[0].u_(0), tc[0].u_(1), tc[0].snormal_(0), tc[0].cov_(0, 0), tc[0].cov_(0, 1),
                ok ? 1 : 0, cfear_tx(T1[2]), cfear_ty(T1[2]), cfear_yaw(T1[2]), reg.itr_ ? reg.itr_ : 0, 0.0, cov[2](0, 0), cov_scale, has_scale ? 1 : 0,
                cost_ok ? 1 : 0, score, residuals.size(), residuals.empty() ? 0.0 : residuals[0], score_after_get_cost,
                ok_soft ? 1 : 0, cfear_tx(T2[2]), cfear_ty(T2[2]), cfear_yaw(T2[2]), reg.summary_.num_residuals,
                ok_other ? 1 : 0, cfear_tx(To[2]), cfear_ty(To[2]), cfear_yaw(To[2]), other.summary_.num_residuals, coarse->GetSize(),
                raw->GetSize(), npts[2], raw_nn.empty() ? -1 : raw_nn[0], raw_cells[5].scale_, raw_cells[5].cov_(0, 0), moved->GetSize(), moved_nn.empty() ? -1 : moved_nn[0],
                tw_same, tw_mutated, tw_mutated_differs ? 1 : 0, tw_sibling_mutated, tw_sibling_other_motion, tw_map, tw_map_differs ? 1 : 0,
                (double)ca->points[0].x, (double)ca->points[0].y);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
  return 0;
}
