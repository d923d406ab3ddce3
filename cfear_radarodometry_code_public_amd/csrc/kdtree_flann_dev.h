// kdtree_flann_dev.h -- PARITY MODE ONLY (cfear_tune NN_TIE_RULE = 2): the 1-NN search of MapPointNormal::GetClosestIdx
// (pointnormal.cpp:238-254) the way the reference's own library answers it, so that exact-distance ties fall as they do there.
//
// GetClosestIdx calls pcl::KdTreeFLANN<pcl::PointXY>::nearestKSearch(p, 1): flann::KDTreeSingleIndex<L2_Simple<float>> built with
// KDTreeSingleIndexParams(15) (leaf_max_size 15, reorder) over the float cell means (ComputeSearchTreeFromCells, pointnormal.cpp:151-162)
// and searched with eps 0 into a KNNSimpleResultSet of one. A point replaces the best so far only if STRICTLY nearer, so of several
// exactly equidistant cells - 1-5 % of a scan's cells share their float mean with another cell - the first one the best-child-first
// descent visits is returned: a property of the tree's layout and of the side of the splitting planes the query lies on, not of the
// cell index. The production search (a uniform grid, lowest index on ties) cannot know that order; this header builds the same tree
// (divideTree / middleSplit_ / planeSplit, restated from the published algorithm, FLANN 1.8 / 1.9 kdtree_single_index.h - the library
// is not under /root/reference) and descends it the same way. One thread builds (a few hundred cells, ~100-200 us: nothing a production
// path could afford per sweep); every query walks the tree with an explicit stack in memory. Same arithmetic, in float, as
// oracle/cfear_oracle.c kd_* (CFO_PERT_NN_TIE_FLANN), which the tests compare with.
#pragma once

namespace cfear_dev {

struct KdNode { int left, right, divfeat, child1, child2; float divlow, divhigh; int pad; };  // child1 < 0: leaf over vind[left, right)
struct KdInterval { float low, high; };
struct KdTree {  // part of ScanDev (arrays in the scan's block; null when the mode is off)
  KdNode* nodes;   // [2 * cap_cells + 2]
  int* vind;       // [cap_cells] the permutation divideTree leaves behind
  float* data;     // [cap_cells][2] reordered copy: point vind[i] at row i (reorder_)
  float bbox[4];   // root bounding box: x low, x high, y low, y high
  int root, n;
};
struct KdFrame {  // one activation of divideTree
  int node, left, right, state, idx, cutfeat, c1, pad;
  KdInterval bbox[2], lb[2], rb[2];
};

__device__ inline void kd_minmax(const float* pts, const int* ind, int count, int dim, float* mn, float* mx) {  // computeMinMax
  *mn = pts[2 * ind[0] + dim]; *mx = *mn;
  for (int i = 1; i < count; i++) {
    const float v = pts[2 * ind[i] + dim];
    if (v < *mn) *mn = v;
    if (v > *mx) *mx = v;
  }
}
__device__ inline void kd_plane_split(const float* pts, int* ind, int count, int cutfeat, float cutval, int* lim1, int* lim2) {  // planeSplit
  int left = 0, right = count - 1;
  for (;;) {
    while (left <= right && pts[2 * ind[left] + cutfeat] < cutval) ++left;
    while (left <= right && pts[2 * ind[right] + cutfeat] >= cutval) --right;
    if (left > right) break;
    { const int x = ind[left]; ind[left] = ind[right]; ind[right] = x; } ++left; --right;
  }
  *lim1 = left;
  right = count - 1;
  for (;;) {
    while (left <= right && pts[2 * ind[left] + cutfeat] <= cutval) ++left;
    while (left <= right && pts[2 * ind[right] + cutfeat] > cutval) --right;
    if (left > right) break;
    { const int x = ind[left]; ind[left] = ind[right]; ind[right] = x; } ++left; --right;
  }
  *lim2 = left;
}
__device__ inline void kd_middle_split(const float* pts, int* ind, int count, int* index, int* cutfeat, float* cutval, const KdInterval* bbox) {  // middleSplit_
  const float EPS = 0.00001f;
  float max_span = bbox[0].high - bbox[0].low;
  { const float span = bbox[1].high - bbox[1].low; if (span > max_span) max_span = span; }
  float max_spread = -1;
  *cutfeat = 0;
  for (int i = 0; i < 2; i++) {
    const float span = bbox[i].high - bbox[i].low;
    if (span > (float)((1 - EPS) * max_span)) {
      float mn, mx;
      kd_minmax(pts, ind, count, i, &mn, &mx);
      const float spread = (float)(mx - mn);
      if (spread > max_spread) { *cutfeat = i; max_spread = spread; }
    }
  }
  const float split_val = (bbox[*cutfeat].low + bbox[*cutfeat].high) / 2;
  float mn, mx;
  kd_minmax(pts, ind, count, *cutfeat, &mn, &mx);
  if (split_val < mn) *cutval = mn;
  else if (split_val > mx) *cutval = mx;
  else *cutval = split_val;
  int lim1, lim2;
  kd_plane_split(pts, ind, count, *cutfeat, *cutval, &lim1, &lim2);
  if (lim1 > count / 2) *index = lim1;
  else if (lim2 < count / 2) *index = lim2;
  else *index = count / 2;
}

// buildIndex by ONE thread: divideTree's recursion unrolled over an explicit stack of activations (`frames`, max_frames of them in
// memory). pts: the n float cell means (x, y). Returns false when the stack would overflow (a degenerate cloud; the caller reports it).
__device__ inline bool kd_build_serial(KdTree* T, const float* pts, int n, KdFrame* frames, int max_frames) {
  T->n = n; T->root = -1;
  if (n <= 0) return true;
  for (int i = 0; i < n; i++) T->vind[i] = i;
  KdInterval rb[2];
  for (int i = 0; i < 2; i++) rb[i].low = rb[i].high = pts[i];  // computeBoundingBox
  for (int k = 1; k < n; k++)
    for (int i = 0; i < 2; i++) {
      if (pts[2 * k + i] < rb[i].low) rb[i].low = pts[2 * k + i];
      if (pts[2 * k + i] > rb[i].high) rb[i].high = pts[2 * k + i];
    }
  int nnodes = 0, sp = 0;
  if (max_frames < 1) return false;
  frames[0].node = nnodes++; frames[0].left = 0; frames[0].right = n; frames[0].state = 0; frames[0].bbox[0] = rb[0]; frames[0].bbox[1] = rb[1];
  KdInterval ret[2];  // the bounding box an activation hands back (divideTree's by-reference argument)
  ret[0] = rb[0]; ret[1] = rb[1];
  while (sp >= 0) {
    KdFrame* f = &frames[sp];
    KdNode* nd = &T->nodes[f->node];
    if (f->state == 0) {
      if (f->right - f->left <= 15) {  // leaf_max_size_: a leaf, its bounding box from its points
        nd->child1 = nd->child2 = -1; nd->left = f->left; nd->right = f->right;
        KdInterval b[2];
        for (int i = 0; i < 2; i++) b[i].low = b[i].high = pts[2 * T->vind[f->left] + i];
        for (int k = f->left + 1; k < f->right; k++)
          for (int i = 0; i < 2; i++) {
            const float v = pts[2 * T->vind[k] + i];
            if (b[i].low > v) b[i].low = v;
            if (b[i].high < v) b[i].high = v;
          }
        ret[0] = b[0]; ret[1] = b[1];
        sp--;
        continue;
      }
      int idx, cutfeat; float cutval;
      kd_middle_split(pts, T->vind + f->left, f->right - f->left, &idx, &cutfeat, &cutval, f->bbox);
      nd->divfeat = cutfeat;
      f->idx = idx; f->cutfeat = cutfeat;
      f->lb[0] = f->bbox[0]; f->lb[1] = f->bbox[1]; f->lb[cutfeat].high = cutval;
      f->rb[0] = f->bbox[0]; f->rb[1] = f->bbox[1]; f->rb[cutfeat].low = cutval;
      f->state = 1;
      if (sp + 1 >= max_frames) return false;
      KdFrame* g = &frames[++sp];  // child1 = divideTree(left, left + idx, left_bbox)
      g->node = nnodes++; g->left = f->left; g->right = f->left + idx; g->state = 0; g->bbox[0] = f->lb[0]; g->bbox[1] = f->lb[1];
      f->c1 = g->node;
    } else if (f->state == 1) {
      f->lb[0] = ret[0]; f->lb[1] = ret[1];  // left_bbox as the child left it
      f->state = 2;
      KdFrame* g = &frames[++sp];  // child2 = divideTree(left + idx, right, right_bbox)   (sp + 1 < max_frames: checked above)
      g->node = nnodes++; g->left = f->left + f->idx; g->right = f->right; g->state = 0; g->bbox[0] = f->rb[0]; g->bbox[1] = f->rb[1];
      nd->child1 = f->c1; nd->child2 = g->node;
    } else {
      f->rb[0] = ret[0]; f->rb[1] = ret[1];
      nd->divlow = f->lb[f->cutfeat].high; nd->divhigh = f->rb[f->cutfeat].low;
      for (int i = 0; i < 2; i++) {
        ret[i].low = f->lb[i].low < f->rb[i].low ? f->lb[i].low : f->rb[i].low;
        ret[i].high = f->lb[i].high > f->rb[i].high ? f->lb[i].high : f->rb[i].high;
      }
      sp--;
    }
  }
  T->bbox[0] = ret[0].low; T->bbox[1] = ret[0].high; T->bbox[2] = ret[1].low; T->bbox[3] = ret[1].high;
  for (int i = 0; i < n; i++) { T->data[2 * i] = pts[2 * T->vind[i]]; T->data[2 * i + 1] = pts[2 * T->vind[i] + 1]; }
  T->root = 0;
  return true;
}

// findNeighbors + searchLevel (eps 0, KNNSimpleResultSet of one): nearest cell of (qx, qy) and its squared distance; -1 for an empty
// tree. stack: this thread's own KD_STACK entries in memory (the sibling subtrees to visit, with the lower bound and the per-dimension
// distances searchLevel would have passed to them; their test against the best so far is made when they are taken up - after the
// nearer subtree has been searched, as in the recursion).
struct KdVisit { int node; float mindistsq, d0, d1; };
#define CFEAR_KD_STACK 64
__device__ inline int kd_nearest(const KdTree* T, float qx, float qy, KdVisit* stack, float* dist_out) {
  if (T->n <= 0 || T->root < 0) return -1;
  const float vec[2] = {qx, qy};
  float d0 = 0, d1 = 0, distsq = 0;  // computeInitialDistances
  if (vec[0] < T->bbox[0]) { d0 = (vec[0] - T->bbox[0]) * (vec[0] - T->bbox[0]); distsq += d0; }
  if (vec[0] > T->bbox[1]) { d0 = (vec[0] - T->bbox[1]) * (vec[0] - T->bbox[1]); distsq += d0; }
  if (vec[1] < T->bbox[2]) { d1 = (vec[1] - T->bbox[2]) * (vec[1] - T->bbox[2]); distsq += d1; }
  if (vec[1] > T->bbox[3]) { d1 = (vec[1] - T->bbox[3]) * (vec[1] - T->bbox[3]); distsq += d1; }
  float worst = 3.402823466e38f;
  int best = -1, sp = 0;
  stack[0].node = T->root; stack[0].mindistsq = distsq; stack[0].d0 = d0; stack[0].d1 = d1;
  bool first = true;
  while (sp >= 0) {
    KdVisit v = stack[sp--];
    if (!first && !(v.mindistsq * 1.0f <= worst)) continue;  // `if (mindistsq * epsError <= result_set.worstDist()) searchLevel(otherChild ...)`
    first = false;
    int node = v.node;
    float mind = v.mindistsq, dd[2] = {v.d0, v.d1};
    for (;;) {  // down the nearer children to a leaf, leaving the farther ones on the stack
      const KdNode nd = T->nodes[node];
      if (nd.child1 < 0) {
        const float worst_dist = worst;
        for (int i = nd.left; i < nd.right; ++i) {
          float result = 0, diff;  // L2_Simple
          diff = vec[0] - T->data[2 * i]; result += diff * diff;
          diff = vec[1] - T->data[2 * i + 1]; result += diff * diff;
          if (result < worst_dist && !(result >= worst)) { worst = result; best = T->vind[i]; }  // addPoint: `if (dist >= worst_distance_) return;`
        }
        break;
      }
      const int idx = nd.divfeat;
      const float val = vec[idx];
      const float diff1 = val - nd.divlow, diff2 = val - nd.divhigh;
      int bestc, other; float cut_dist;
      if ((diff1 + diff2) < 0) { bestc = nd.child1; other = nd.child2; cut_dist = (val - nd.divhigh) * (val - nd.divhigh); }
      else { bestc = nd.child2; other = nd.child1; cut_dist = (val - nd.divlow) * (val - nd.divlow); }
      if (sp + 1 < CFEAR_KD_STACK) {
        KdVisit o;
        o.node = other; o.mindistsq = mind + cut_dist - dd[idx];
        o.d0 = idx == 0 ? cut_dist : dd[0]; o.d1 = idx == 1 ? cut_dist : dd[1];
        stack[++sp] = o;
      }
      node = bestc;  // searchLevel(bestChild, mindistsq, dists)
    }
  }
  *dist_out = worst;
  return best;
}

}  // namespace cfear_dev
