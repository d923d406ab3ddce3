#!/bin/bash
# per-phase instruction counts of the k-strongest kernel (debug early-outs), run on the GPU box
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for d in 1 2 3 0; do
  K1_REPS=1 K1_CONFIGS="8,2,$d" rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES -d $R/gpurun_out/pmc_k1_dbg$d -o k1 -- python $R/tools/gpu_time_k1.py > $R/gpurun_out/pmc_k1_dbg$d.log 2>&1
done
