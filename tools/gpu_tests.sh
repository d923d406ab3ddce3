set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -40) > gpurun_out/gpu_tests.log 2>&1
tail -15 gpurun_out/gpu_tests.log
