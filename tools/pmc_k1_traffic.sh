#!/bin/bash
# HBM traffic of the k-strongest kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# (MI355X_MICROARCH.md "HBM": TCC slots do not fit both; FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  K1_REPS=2 K1_CONFIGS="8,2,0" rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_k1_$c -o k1 -- python $R/tools/gpu_time_k1.py > $R/gpurun_out/pmc_k1_$c.log 2>&1
done
cd $R && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
tail -1 $R/gpurun_out/prof_bench.log | cut -c1-300
