// register_step_large.hip -- the batched registration step kernel for LARGE submaps (submap_scan_size 8 ... 63: the reference's s10 presets,
// params/baseline_p2d/oxford_cfear-3-s10, and CFEAR-3-s50, launch/oxford_demo:62-71 - its most accurate published setting).
//
// A registration against fifty keyframes is 13 groups of four keyframes to associate (n_scan_normal.cpp:359-367: one scan pair per
// keyframe) and ~6000 residual blocks to evaluate ~45 times. With the production shape - 256 threads, three workgroups per compute unit,
// a 50 KB LDS match array - that was a 2.4 ms chain per registration even alone on a unit: group after group of dependent memory round
// trips with 172 of 256 threads busy, and four fifths of every evaluation streamed from memory. This instantiation gives a registration a
// compute unit to itself, as replay.hip does for a whole sweep: 512 threads (the (group, cell) items of an association are dealt to
// them densely: 5 passes instead of 13), ALL EIGHT waves evaluate, and the unit's whole LDS holds the matches (2250 residual blocks
// with all eight arrays; P2L 2571, P2P 3600). Same device code, same results as the 256-thread kernels up to the summation order of
// the evaluation's partial sums (eight waves instead of four).
#ifndef CFEAR_LARGE_BLOCK
#define CFEAR_LARGE_BLOCK 512    // tools: A/B builds (256 with two workgroups per unit: the 512-register budget per thread, two controllers that take turns)
#endif
#define CFEAR_REG_BLOCK CFEAR_LARGE_BLOCK
#define CFEAR_EVAL_WAVES (CFEAR_LARGE_BLOCK / 64)
#ifndef CFEAR_LARGE_WG_PER_CU
#define CFEAR_LARGE_WG_PER_CU 1  // tools: A/B builds (2: half the unit's LDS each)
#endif
#if CFEAR_LARGE_WG_PER_CU == 1
#define CFEAR_MATCH_LDS_CAP 2250
#define CFEAR_REG_LDS_BUDGET (160 * 1024)
#else
#define CFEAR_MATCH_LDS_CAP 1040
#define CFEAR_REG_LDS_BUDGET (80 * 1024)
#endif
#include "common.h"
#include "odometry_step_dev.h"

namespace {
template <bool TIMED, int KCOST>
__global__ __launch_bounds__(BLOCK_R, (CFEAR_LARGE_BLOCK / 256) * CFEAR_LARGE_WG_PER_CU /* waves per SIMD */) void register_step_large_kernel(OdoParams OP, SeqState* states, const BlockScratch* scratch, double* cov_work,
                                                                      cfear_reg_summary* summaries, double* poses_out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[RegLds::total];
  register_step_body<TIMED, KCOST>(lds, OP.order ? OP.order[blockIdx.x] : OP.seq0 + (int)blockIdx.x, OP, states, scratch, cov_work, summaries, poses_out);
}
}  // namespace

// pipeline.hip (launch_register_step): one instantiation per cost metric, the per-phase timers with the cost read at run time
__attribute__((visibility("hidden"))) void cfear_launch_register_step_large(const void* odo_params, int count, hipStream_t st, void* states, const void* scratch,
                                                                           double* cov_work, cfear_reg_summary* summaries, double* poses_out) {
  const OdoParams& P = *static_cast<const OdoParams*>(odo_params);
#define CFEAR_LAUNCH_REG(T, C) hipLaunchKernelGGL((register_step_large_kernel<T, C>), dim3(count), dim3(BLOCK_R), 0, st, P, static_cast<SeqState*>(states), \
                                                  static_cast<const BlockScratch*>(scratch), cov_work, summaries, poses_out)
  if (P.phase_times) CFEAR_LAUNCH_REG(true, -1);
  else if (P.rp.cost == CFEAR_COST_P2L) CFEAR_LAUNCH_REG(false, CFEAR_COST_P2L);
  else if (P.rp.cost == CFEAR_COST_P2D) CFEAR_LAUNCH_REG(false, CFEAR_COST_P2D);
  else CFEAR_LAUNCH_REG(false, CFEAR_COST_P2P);
#undef CFEAR_LAUNCH_REG
}
