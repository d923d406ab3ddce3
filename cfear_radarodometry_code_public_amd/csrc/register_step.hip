// register_step.hip -- the batched registration step kernel as it runs in production: compiled for registrations of at most
// CFEAR_STEP_SMALL_SCANS scans (submap_scan_size <= 7: the reference's presets use 3 and 4). The per-scan arrays of the workgroup's
// shared state shrink from 64 entries to 8, and the ~10 KB of LDS that frees go to the match array: 784 residual blocks with all
// eight arrays instead of 622 (P2L, seven arrays: 896 instead of 710; P2P, five: 1254 instead of 995) at the same three workgroups
// per compute unit. A city-block scene at k = 12 builds 650-750 blocks per registration: with 710 in LDS two out of five
// registrations evaluated part of their matches from memory. pipeline.hip keeps an instantiation for larger submaps.
#define CFEAR_REG_MAX_SCANS 8
#ifndef CFEAR_MATCH_LDS_CAP  // (tools/reg_resources.sh, tools/build_variant.sh: A/B builds)
#define CFEAR_MATCH_LDS_CAP 784
#endif
#include "common.h"
#include "odometry_step_dev.h"

static_assert(CFEAR_REG_MAX_SCANS == CFEAR_STEP_SMALL_SCANS, "common.h tells pipeline.hip when to launch these kernels");

// pipeline.hip (launch_register_step): one instantiation per cost metric, the per-phase timers with the cost read at run time
__attribute__((visibility("hidden"))) void cfear_launch_register_step_small(const void* odo_params, int count, hipStream_t st, void* states, void* const* scan_slots,
                                                                           const void* scratch, double* poses_work, double* cov_work,
                                                                           cfear_reg_summary* summaries, double* poses_out) {
  const OdoParams& P = *static_cast<const OdoParams*>(odo_params);
#define CFEAR_LAUNCH_REG(T, C) hipLaunchKernelGGL((register_step_kernel<T, C>), dim3(count), dim3(BLOCK_R), 0, st, P, static_cast<SeqState*>(states), \
                                                  reinterpret_cast<ScanDev* const*>(scan_slots), static_cast<const BlockScratch*>(scratch), poses_work, cov_work, summaries, poses_out)
  if (P.phase_times) CFEAR_LAUNCH_REG(true, -1);
  else if (P.rp.cost == CFEAR_COST_P2L) CFEAR_LAUNCH_REG(false, CFEAR_COST_P2L);
  else if (P.rp.cost == CFEAR_COST_P2D) CFEAR_LAUNCH_REG(false, CFEAR_COST_P2D);
  else CFEAR_LAUNCH_REG(false, CFEAR_COST_P2P);
#undef CFEAR_LAUNCH_REG
}
