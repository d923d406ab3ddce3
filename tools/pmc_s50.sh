#!/bin/bash
# memory-side counters of the registration kernel of the CFEAR-3-s50 preset (768 sequences), both kernel shapes; one counter per pass
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for shape in 1 2; do
  for grp in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; do
    rm -rf /tmp/pmc_s50
    CFEAR_PRESET_LARGE_KERNEL=$shape CFEAR_BENCH_PRESETS=cfear3_s50 timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_s50 -o s50 -- python $R/tools/gpu_presets.py > $R/gpurun_out/pmc_s50_${shape}_$grp.log 2>&1
    echo "shape $shape $grp: $(grep -h 'cfear3_s50' $R/gpurun_out/pmc_s50_${shape}_$grp.log | tail -1)"
    (cd $R; ROCPD_LAST=12 python tools/rocpd_summary.py $(find /tmp/pmc_s50 -name "*.db" | head -1) 2>/dev/null | grep -E "register_step[a-z_0-9]*kernel<false" | grep -E "\| [A-Z]")
  done
done
