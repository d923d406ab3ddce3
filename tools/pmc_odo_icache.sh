#!/bin/bash
# instruction-cache counters of the features / registration step kernels (1024 sequences, single stream)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  ODO_FRAMES=14 ODO_CFG="0,1536" timeout 150 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_odo_ic$i -o odo -- python $R/tools/gpu_odo_streams.py > $R/gpurun_out/pmc_odo_ic$i.log 2>&1
  echo "== $grp (rc $?)"
  (cd $R; python tools/rocpd_summary.py $(find /tmp/pmc_odo_ic$i -name "*.db" | head -1) 2>&1 | grep -E "step_kernel<false> \| [A-Za-z_]+" )
done
