// eval_kitti -- drift of an estimated trajectory against ground truth, both in the KITTI text format the offline
// harness writes (est_00.txt / gt_00.txt under --est_directory / --gt_directory in the reference's runs).
// usage: eval_kitti <gt.txt> <est.txt>
#include <cstdio>

#include "cfear_hip/kitti_metric.hpp"

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <gt.txt> <est.txt>\n", argv[0]); return 2; }
  std::vector<cfear_host::Pose34> gt, est;
  if (!cfear_host::read_kitti_poses(argv[1], gt)) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  if (!cfear_host::read_kitti_poses(argv[2], est)) { std::fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
  const cfear_host::KittiDrift d = cfear_host::kitti_drift(gt, est);
  std::printf("{\"poses_gt\": %zu, \"poses_est\": %zu, \"segments\": %d, \"translation_percent\": %.6f, \"rotation_deg_per_100m\": %.6f}\n",
              gt.size(), est.size(), d.segments, d.translation_percent, d.rotation_deg_per_100m);
  return 0;
}
