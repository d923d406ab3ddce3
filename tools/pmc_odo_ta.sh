#!/bin/bash
# texture-addresser / L1 counters of the features / registration step kernels (1024 sequences, single stream); one small
# counter group per pass, each pass under its own timeout
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  ODO_FRAMES=14 ODO_CFG="0,1536" timeout 150 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_odo_ta$i -o odo -- python $R/tools/gpu_odo_streams.py > $R/gpurun_out/pmc_odo_ta$i.log 2>&1
  echo "== $grp (rc $?)"
  (cd $R; python tools/rocpd_summary.py $(find /tmp/pmc_odo_ta$i -name "*.db" | head -1) 2>&1 | grep -E "step_kernel<false> \| [A-Za-z_]+" )
done
