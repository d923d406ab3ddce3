set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_pipeline_gpu.py -q -m gpu -k "ahead or many_resident or profile" 2>&1 | tail -4) > gpurun_out/r2_tests8.log 2>&1
for n in 0 1 2 3 4 6; do (CFEAR_BENCH_OVERLAP=$n timeout 600 python bench.py --no-cpu-baseline --single-sequence-sweeps 0 --stream-steps 0 --no-isolated 2>&1 | tail -1) > gpurun_out/r2_bench8_$n.log 2>&1; done
