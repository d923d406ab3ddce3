#!/bin/bash
# run on the GPU box: regenerates the files that get copied into profiles/ (prefix = $1, e.g. r01)
P=${1:-r01}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  K1_N=1536 K1_REPS=2 K1_CONFIGS="7,4" timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_k1_$c -o k1 -- python $R/tools/gpu_time_k1.py > $O/${P}_pmc_k1_$c.log 2>&1
done
python $R/tools/pmc_k1_traffic_report.py $(find /tmp/pmc_k1_FETCH_SIZE -name "*.db" | head -1) $(find /tmp/pmc_k1_WRITE_SIZE -name "*.db" | head -1) 1536 $O/${P}_k1_traffic.json
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --stream-steps 0 --single-sequence-sweeps 0 --no-isolated --no-presets > $O/${P}_bench_prof.log 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_bench -name "*.db" | head -1) $O/${P}_bench_kernel_stats.md | head -8
sed -i '1i rocprofv3 --kernel-trace --stats -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --stream-steps 0 --single-sequence-sweeps 0 --no-isolated --no-presets (the command the driver runs, without the legs after the timed regions: 5 repeats x (8 pre-roll + 5 warm-up + 20 timed) = 165 launches of each kernel, 4608 sequences each)\n' $O/${P}_bench_kernel_stats.md
cd $R && timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${P}_bench.json 2> $O/${P}_bench.err; tail -c 600 $O/${P}_bench.json
# instruction-mix counters of the step kernels (own pass: --pmc with --kernel-trace only)
cd /tmp; ODO_FRAMES=14 ODO_CFG="0,1536" timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES -d /tmp/pmc_odo_r -o odo -- python $R/tools/gpu_odo_streams.py > $O/${P}_pmc_odo.log 2>&1
cd $R; python tools/rocpd_summary.py $(find /tmp/pmc_odo_r -name "*.db" | head -1) | grep -E "step_kernel|kstrongest" > $O/${P}_odo_pmc.txt
# HBM traffic of the step kernels (one counter per pass) -> ${P}_odo_mem.txt
(bash $R/tools/pmc_odo_mem.sh) > $O/${P}_odo_mem.txt 2>&1
# the k sweep (general cloud / feature paths beyond k = 12) and the driving-like replay parity with its replay rate
cd $R; timeout 600 python tools/gpu_k_sweep.py > $O/${P}_k_sweep.txt 2>&1
for k in blocks canyon field; do timeout 600 python tests/run_drive_parity.py $k ${DRIVE_SWEEPS:-2000} $O/${P}_drive_$k.json > /dev/null 2>&1; done
# round 4: the filter by input family (S-uniform / S-world / S-ties) and by phase (stop-variant builds: tools/build_k1_stop_variants.sh, run before gpurun),
# the large-submap presets on long drives (every sweep against the oracle), the launch-order A/B
bash $R/tools/pmc_k1_inputs.sh $O/${P}_k1_inputs.txt > /dev/null 2>&1
[ -f $R/tools/_stop/libcfear_hip_k1stop1.so ] && bash $R/tools/pmc_k1_phases.sh $O/${P}_k1_phases.txt > /dev/null 2>&1
cd $R; for ps in s10_p2p s10_p2d s50_cfear3; do timeout 900 python tests/run_drive_parity.py canyon ${DRIVE_SWEEPS_LARGE:-2000} $O/${P}_drive_canyon_$ps.json $ps > /dev/null 2>&1; done
bash $R/tools/ab_reg_order.sh $O/${P}_ab_reg_order.txt > /dev/null 2>&1
# round 5: the large-submap registration kernel's counters after the instruction diet, the CA-CFAR detector alone, the whole GPU test suite
(bash $R/tools/pmc_s50_sq.sh) > $O/${P}_s50_sq.txt 2>&1
(bash $R/tools/pmc_s50.sh) > $O/${P}_odo_mem_s50_after.txt 2>&1
cd $R; timeout 300 python tools/gpu_time_cfar.py 2>&1 | grep -v amdgpu > $O/${P}_cfar_kernel_final.txt
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12) > $O/${P}_gpu_tests.log 2>&1
for ps in cfear1 nocomp_p2p_k40 res1_s3; do timeout 900 python tests/run_drive_parity.py canyon ${DRIVE_SWEEPS_LARGE:-2000} $O/${P}_drive_canyon_$ps.json $ps > /dev/null 2>&1; done
timeout 900 python tests/run_drive_parity.py blocks 10000 $O/${P}_drive10k_blocks_cfear1.json cfear1 > /dev/null 2>&1
for k in canyon blocks; do timeout 900 python tests/run_drive_parity.py $k ${DRIVE_SWEEPS_LARGE:-2000} $O/${P}_drive_${k}_ca_cfar.json ca_cfar > /dev/null 2>&1; done  # (the oracle's literal detector: ~75 ms per sweep)
# round 6: the street world (reflectivity fixed to the surfaces) with CFEAR-3 as shipped and the large-submap presets; the drop-in route; the CA-CFAR detector's counters
for ps in cfear3_k40_p2p s10_p2p s50_cfear3; do timeout 900 python tests/run_drive_parity.py street ${DRIVE_SWEEPS_LARGE:-2000} $O/${P}_drive_street_$ps.json $ps > /dev/null 2>&1; done
timeout 900 python tests/run_drive_parity.py street ${DRIVE_SWEEPS:-2000} $O/${P}_drive_street.json > /dev/null 2>&1
timeout 300 python tools/gpu_dropin.py 2000 > $O/${P}_dropin_phases.txt 2>&1
(bash $R/tools/pmc_cfar.sh) > $O/${P}_cfar_pmc.txt 2>&1
(bash $R/tools/gpu_time_k1_pair.sh) > /dev/null 2>&1   # -> ${O}/r06_k1_pair.txt (the two-rows-at-once filter variant against the production kernel)
