#!/bin/bash
# A/B of the large-submap registration shapes on the s10 / s50 preset legs: the product library (512 threads, one workgroup per unit, from 24 keyframes
# or for few sequences; 256 threads x 3 per unit below) against tools/_stop/libcfear_hip_med.so = tools/build_variant.sh med "-DCFEAR_LARGE_BLOCK=256
# -DCFEAR_LARGE_WG_PER_CU=2" (256 threads, two workgroups per unit, 80 KB of LDS and up to 256 registers each)
cd $GRAFT_REPO_ROOT
run() { n=$1; shift; echo "== $n"; env "$@" CFEAR_BENCH_PRESETS=cfear3_s50,cfear3_s10_p2p,cfear3_s10_p2d python tools/gpu_presets.py 2>&1 | grep -v amdgpu; }
run "product library, its own launch policy" X=1
run "product library, register_step_large.hip forced (512 x 1)" CFEAR_PRESET_LARGE_KERNEL=2
run "variant 256 x 2, the policy (s50 only)" CFEAR_HIP_LIB=tools/_stop/libcfear_hip_med.so
run "variant 256 x 2 forced" CFEAR_HIP_LIB=tools/_stop/libcfear_hip_med.so CFEAR_PRESET_LARGE_KERNEL=2
