#!/bin/bash
# round 6: the two-rows-at-once variant of the k-strongest kernel (kstrongest_pair_kernel, CFEAR_K1_PAIR=1) against the production kernel:
# bit-exactness (the filter's test file under the switch), time per launch on the three input families, instructions per row, and the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
{
echo "== tests/test_kstrongest_gpu.py with CFEAR_K1_PAIR=1"
CFEAR_K1_PAIR=1 timeout 900 python -m pytest tests/test_kstrongest_gpu.py tests/test_golden_gpu.py -x -q 2>&1 | tail -3
for rows in 4 6 8 12; do
  echo "== production kernel, $rows rows per wave, 1536-scan launches"; K1_TIES=1 K1_UNIFORM_SEEDED=1 K1_N=1536 K1_REPS=3 K1_CONFIGS="7,$rows" timeout 300 python tools/gpu_time_k1.py 2>&1 | grep occ=
  echo "== pair kernel, $rows rows per wave";                            CFEAR_K1_PAIR=1 K1_TIES=1 K1_UNIFORM_SEEDED=1 K1_N=1536 K1_REPS=3 K1_CONFIGS="7,$rows" timeout 300 python tools/gpu_time_k1.py 2>&1 | grep occ=
done
echo "== instructions per row (uniform, world), production then pair"
bash tools/pmc_k1_valu.sh "7,6" 2>&1 | tail -3
CFEAR_K1_PAIR=1 bash tools/pmc_k1_valu.sh "7,6" 2>&1 | tail -3
echo "== the step (bench.py --no-presets ...), production then pair"
for pr in 0 1; do CFEAR_K1_PAIR=$pr timeout 600 python bench.py --no-cpu-baseline --no-presets --stream-steps 0 --single-sequence-sweeps 0 --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pair=$pr scans/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4), 'filter us', round(d['kernels']['kstrongest_launch_us'],1), 'frac', round(d['roofline']['frac'],3))"; done
} > $O/r06_k1_pair.txt 2>&1
cat $O/r06_k1_pair.txt
