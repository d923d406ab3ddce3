set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do (timeout 600 python bench.py --no-cpu-baseline --single-sequence-sweeps 0 --stream-steps 0 --steps 40 2>&1 | tail -1) > gpurun_out/r2_bench6_$i.log 2>&1; done
