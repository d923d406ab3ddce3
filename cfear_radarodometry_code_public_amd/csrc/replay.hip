// replay.hip -- a chunk of consecutive sweeps of every sequence in ONE launch: a persistent 512-thread workgroup per sequence
// runs features -> registration -> keyframe logic sweep after sweep (offline_odometry.cpp:103-125's loop body,
// odometrykeyframefuser.cpp:143-259) from the slots the batched filter left for the whole chunk.
//
// Why: within a sequence the sweeps are strictly sequential (the motion compensation of sweep t uses the motion estimated at
// t - 1), so a replay of few sequences is a chain of two dependent kernels per sweep - ~45 + ~105 us of work and, launched one
// by one, two launch gaps of ~10 us on top. Here nothing is launched in between: the state stays where it is, the only
// synchronisation between the two stages is a workgroup barrier. The registration code is compiled for this workgroup size
// (CFEAR_REG_BLOCK = 512: the association takes 512 source cells per pass, waves 4..7 sit out the evaluations), the LDS
// segment is the larger of the two stages' (one workgroup per compute unit is plenty for a replay: there are few sequences).
#define CFEAR_REG_BLOCK 512
// one workgroup per compute unit: the LDS the feature stage does not need while the registration runs is not the limit, the unit's
// 160 KB are - 1200 residual blocks with all eight arrays (P2L 1371, P2P 1920) stay in LDS, a dense street canyon's 1100-1200 included
#define CFEAR_MATCH_LDS_CAP 1200
#define CFEAR_REG_LDS_BUDGET (160 * 1024)
#include "common.h"
#include "odometry_step_dev.h"

namespace {
constexpr size_t kChunkLds = FeatLdsC::total > RegLds::total ? FeatLdsC::total : RegLds::total;

// the two stages out of line: each gets a register allocation of its own (inlined into the loop, the allocator ran out of the
// 256 registers a 512-thread workgroup can have and spilled 158 of them)
__device__ __noinline__ void features_stage(unsigned char* lds, int q, const uint32_t* slots, const double* trig, const OdoParams* P, const SeqState* states,
                                            const BlockScratch* scratch) {
  features_step_body<false>(lds, q, slots, trig, *P, states, scratch);
}
__device__ __noinline__ void features_cloud_stage(unsigned char* lds, int q, const float* xyi, int cap, const int* counts, const OdoParams* P, const SeqState* states,
                                                  const BlockScratch* scratch) {
  features_cloud_step_body(lds, q, xyi, cap, counts, *P, states, scratch);
}
__device__ __noinline__ void register_stage(unsigned char* lds, int q, const OdoParams* P, SeqState* states, const BlockScratch* scratch, double* cov_work,
                                            cfear_reg_summary* summaries, double* poses_out) {
  register_step_body<false>(lds, q, *P, states, scratch, cov_work, summaries, poses_out);
}

__global__ __launch_bounds__(BLOCK_F) void replay_chunk_kernel(const uint32_t* slots_chunk /*[cnt][B][A * k]*/, int cnt, int B, const double* trig,
                                                               OdoParams OP, SeqState* states, const BlockScratch* scratch, double* cov_work,
                                                               cfear_reg_summary* summaries, double* poses_out, cfear_sweep_record* records /*[cnt][B] or null*/) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kChunkLds];
  const int q = OP.seq0 + (int)blockIdx.x;
  const size_t sweep_slots = (size_t)B * OP.A * OP.k;
  __shared__ OdoParams P;  // by pointer to the stages: LDS, not a per-thread stack copy
  if (threadIdx.x == 0) P = OP;
  for (int t = 0; t < cnt; t++) {
    if (threadIdx.x == 0) P.records = records ? records + (size_t)t * B : nullptr;
    __syncthreads();
    features_stage(lds, q, slots_chunk + sweep_slots * (size_t)t, trig, &P, states, scratch);
    __syncthreads();  // the scan is complete (and visible to the whole workgroup) before it is registered
    register_stage(lds, q, &P, states, scratch, cov_work, summaries, poses_out);
    __syncthreads();  // the state of the sequence (motion, keyframe ring, free slot) is written before the next sweep reads it
  }
}
// the same chunk from clouds (filter_type CA-CFAR): [cnt][B][cap][3] floats and [cnt][B] counts
__global__ __launch_bounds__(BLOCK_F) void replay_chunk_cloud_kernel(const float* xyi_chunk, int cap, const int* counts_chunk, int cnt, int B, OdoParams OP, SeqState* states,
                                                                     const BlockScratch* scratch, double* cov_work, cfear_reg_summary* summaries, double* poses_out,
                                                                     cfear_sweep_record* records /*[cnt][B] or null*/) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[kChunkLds];
  const int q = OP.seq0 + (int)blockIdx.x;
  __shared__ OdoParams P;
  if (threadIdx.x == 0) P = OP;
  for (int t = 0; t < cnt; t++) {
    if (threadIdx.x == 0) P.records = records ? records + (size_t)t * B : nullptr;
    __syncthreads();
    features_cloud_stage(lds, q, xyi_chunk + 3 * (size_t)cap * B * t, cap, counts_chunk + (size_t)B * t, &P, states, scratch);
    __syncthreads();
    register_stage(lds, q, &P, states, scratch, cov_work, summaries, poses_out);
    __syncthreads();
  }
}
}  // namespace

__attribute__((visibility("hidden"))) void cfear_launch_replay_chunk_cloud(const float* d_xyi, int cap, const int* d_counts, int cnt, int B, const void* odo_params,
                                                                          void* states, const void* scratch, double* cov_work, cfear_reg_summary* summaries,
                                                                          double* poses_out, cfear_sweep_record* records, hipStream_t stream) {
  const OdoParams& OP = *static_cast<const OdoParams*>(odo_params);
  hipLaunchKernelGGL(replay_chunk_cloud_kernel, dim3(B), dim3(BLOCK_F), 0, stream, d_xyi, cap, d_counts, cnt, B, OP, static_cast<SeqState*>(states),
                     static_cast<const BlockScratch*>(scratch), cov_work, summaries, poses_out, records);
}

// pipeline.hip (cfear_odometry_replay_host): launches the chunk kernel on `stream`
__attribute__((visibility("hidden"))) void cfear_launch_replay_chunk(const uint32_t* d_slots, int cnt, int B, const double* d_trig, const void* odo_params,
                                                                    void* states, const void* scratch, double* cov_work, cfear_reg_summary* summaries,
                                                                    double* poses_out, cfear_sweep_record* records, hipStream_t stream) {
  const OdoParams& OP = *static_cast<const OdoParams*>(odo_params);
  hipLaunchKernelGGL(replay_chunk_kernel, dim3(B), dim3(BLOCK_F), 0, stream, d_slots, cnt, B, d_trig, OP, static_cast<SeqState*>(states),
                     static_cast<const BlockScratch*>(scratch), cov_work, summaries, poses_out, records);
}
