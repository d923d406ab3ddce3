set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/r2_tests2.log 2>&1
tail -5 gpurun_out/r2_tests2.log
