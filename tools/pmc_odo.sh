#!/bin/bash
# instruction-mix counters of the features / registration step kernels (1024 sequences, single stream)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
ODO_FRAMES=14 ODO_CFG="0,1536" rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_odo -o odo -- python $R/tools/gpu_odo_streams.py > $R/gpurun_out/pmc_odo.log 2>&1
cd $R; python tools/rocpd_summary.py $(find /tmp/pmc_odo -name "*.db" | head -1)
