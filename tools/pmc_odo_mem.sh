#!/bin/bash
# memory-side counters of the features / registration step kernels (1024 sequences, single stream); one small
# counter group per pass, each pass under its own timeout
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do  # one counter per pass (together they abort rocprofv3 on this pool)
  i=$((i+1))
  ODO_FRAMES=14 ODO_CFG="0,1536" timeout 150 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_odo_mem$i -o odo -- python $R/tools/gpu_odo_streams.py > $R/gpurun_out/pmc_odo_mem$i.log 2>&1
  (cd $R; python tools/rocpd_summary.py $(find /tmp/pmc_odo_mem$i -name "*.db" | head -1) | grep -E "(step_kernel<false(, -?[0-9]+)?>|kstrongest_kernel<4, 7>) \| [A-Z]")
done
