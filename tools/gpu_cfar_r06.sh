#!/bin/bash
# round 6: the owner-layout CA-CFAR detector - parity tests, then timing of every shape against the round-5 kernels, then per-phase times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1200 python -m pytest tests/test_cfar_gpu.py tests/test_cfar_odometry_gpu.py -x -q 2>&1 | tail -15) > gpurun_out/cfar_tests.log
tail -5 gpurun_out/cfar_tests.log
{
for shp in 28 20 16; do echo "== owner layout, $shp bins per thread"; CFEAR_CFAR_SHAPE=$shp timeout 300 python tools/gpu_time_cfar.py; done
echo "== round-5 kernels"; CFEAR_CFAR_OLD_KERNELS=1 timeout 300 python tools/gpu_time_cfar.py
for shp in 28 16; do for st in 1 2; do echo "== owner layout, $shp bins per thread, trips end after phase $st"; CFAR_PRESET=0 CFEAR_CFAR_STOP=$st CFEAR_CFAR_SHAPE=$shp timeout 300 python tools/gpu_time_cfar.py; done; done
} > gpurun_out/cfar_times.txt 2>&1
grep -v amdgpu.ids gpurun_out/cfar_times.txt
