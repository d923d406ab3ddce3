set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 600 python bench.py --no-cpu-baseline --single-sequence-sweeps 0 --stream-steps 0 --steps 40 2>&1 | tail -1) > gpurun_out/gpu_bench.log 2>&1
(ODO_B=1536 ODO_U=16 timeout 600 python tools/gpu_phase_times.py 2>&1 | tail -12) > gpurun_out/gpu_phases.log 2>&1
