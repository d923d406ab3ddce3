set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for b in 1536 3072 4608 6144; do (timeout 900 python bench.py --steps 20 --warmup 5 --sequences $b --no-cpu-baseline --single-sequence-sweeps 0 --stream-steps 0 --no-isolated 2>&1 | tail -1) > gpurun_out/r2_bench9_$b.log 2>&1; done
