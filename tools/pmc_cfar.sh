#!/bin/bash
# instruction and memory counters of the CA-CFAR detector kernels (1536 sweeps per launch, the reference's preset and two others: tools/gpu_time_cfar.py)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_cfar
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_cfar -o cfar -- python $R/tools/gpu_time_cfar.py > /tmp/pmc_cfar.log 2>&1
  (cd $R; python tools/rocpd_summary.py $(find /tmp/pmc_cfar -name "*.db" | head -1) 2>/dev/null | grep -E "cfar_[a-z_]*kernel")
done
