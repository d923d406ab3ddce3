#!/bin/bash
# instruction and memory counters of the CA-CFAR detector kernels (1536 sweeps per launch; CFAR_PRESET picks one of tools/gpu_time_cfar.py's presets, default the reference's)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
export CFAR_PRESET=${CFAR_PRESET:-0}
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pmc_cfar
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_cfar -o cfar -- python $R/tools/gpu_time_cfar.py > /tmp/pmc_cfar.log 2>&1
  (cd $R; python tools/rocpd_summary.py $(find /tmp/pmc_cfar -name "*.db" | head -1) 2>/dev/null | grep -E "cfar_[a-z_]*kernel")
done
