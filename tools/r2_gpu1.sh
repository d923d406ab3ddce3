set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r2_tests1.log 2>&1
(timeout 600 python bench.py 2>&1 | tail -3) > gpurun_out/r2_bench1.log 2>&1
(CFEAR_BENCH_OVERLAP=0 timeout 600 python bench.py --no-cpu-baseline --single-sequence-sweeps 0 --stream-steps 0 2>&1 | tail -1) > gpurun_out/r2_bench1_nooverlap.log 2>&1
(ODO_B=1536 ODO_U=16 timeout 600 python tools/gpu_phase_times.py 2>&1 | tail -24) > gpurun_out/r2_phase1.log 2>&1
tail -5 gpurun_out/r2_tests1.log
