// stdsort_perm.cpp -- TEST INFRASTRUCTURE ([3P] sensitivity sweep): the permutation PCL <= 1.9's VoxelGrid applies to its points.
// pcl/filters/impl/voxel_grid.hpp builds `std::vector<cloud_point_index_idx> index_vector` in point order and calls
// std::sort(index_vector.begin(), index_vector.end(), std::less<cloud_point_index_idx>()), where operator< compares the voxel
// index only: an UNSTABLE sort, so the order of the points inside a voxel - and with it the last bit of the float centroid - is
// whatever libstdc++'s introsort leaves. This file restates those four lines (a struct, its comparison, one std::sort call)
// so that the oracle can be run with exactly that permutation (oracle/cfear_oracle.c, CFO_PERT_VOXEL_STDSORT).
#include <algorithm>
#include <cstdint>
#include <vector>

namespace {
struct cloud_point_index_idx {
  unsigned int idx;
  unsigned int cloud_point_index;
  bool operator<(const cloud_point_index_idx& p) const { return idx < p.idx; }
};
}  // namespace

extern "C" void cfo_stdsort_perm(uint32_t* voxel_idx, uint32_t* point_idx, int n) {
  std::vector<cloud_point_index_idx> v((size_t)n);
  for (int i = 0; i < n; i++) { v[(size_t)i].idx = voxel_idx[i]; v[(size_t)i].cloud_point_index = point_idx[i]; }
  std::sort(v.begin(), v.end(), std::less<cloud_point_index_idx>());
  for (int i = 0; i < n; i++) { voxel_idx[i] = v[(size_t)i].idx; point_idx[i] = v[(size_t)i].cloud_point_index; }
}
