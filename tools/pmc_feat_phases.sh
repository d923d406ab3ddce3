#!/bin/bash
# per-phase instruction counts of the feature kernel: PMC counters of the stop-after-phase-k builds (tools/build_stop_variants.sh)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/feat_phases.txt; : > $OUT
for k in 1 2 3 4 5 6 7 full; do
  if [ $k = full ]; then unset CFEAR_HIP_LIB; else export CFEAR_HIP_LIB=$R/tools/_stop/libcfear_hip_stop$k.so; fi
  rm -rf /tmp/pmc_fp
  ODO_FRAMES=14 ODO_CFG="0,1536" timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_fp -o fp -- python $R/tools/gpu_odo_streams.py > $R/gpurun_out/feat_phases_$k.log 2>&1
  echo "== stop $k" >> $OUT
  python $R/tools/rocpd_summary.py $(find /tmp/pmc_fp -name "*.db" | head -1) | grep features_step >> $OUT
done
cat $OUT
