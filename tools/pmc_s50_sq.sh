#!/bin/bash
# instruction-side counters of the registration kernel of the CFEAR-3-s50 preset (768 sequences), both kernel shapes
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for shape in 1 2; do
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM" "GRBM_GUI_ACTIVE"; do
    rm -rf /tmp/pmc_s50
    CFEAR_PRESET_LARGE_KERNEL=$shape CFEAR_BENCH_PRESETS=cfear3_s50 timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_s50 -o s50 -- python $R/tools/gpu_presets.py > $R/gpurun_out/pmc_s50_sq.log 2>&1
    echo "shape $shape: $(grep -h 'cfear3_s50' $R/gpurun_out/pmc_s50_sq.log | tail -1)"
    (cd $R; ROCPD_LAST=12 python tools/rocpd_summary.py $(find /tmp/pmc_s50 -name "*.db" | head -1) 2>/dev/null | grep -E "register_step[a-z_0-9]*kernel<false" | grep -E "\| [A-Z]")
  done
done
