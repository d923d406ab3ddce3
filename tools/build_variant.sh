#!/bin/bash
# A/B builds of the library for measurements on one GPU box: tools/build_variant.sh NAME "-DMACRO=1 ..." compiles pipeline.hip and
# replay.hip with the extra flags and links them with the product build's other objects into tools/_stop/libcfear_hip_NAME.so
# (tools/_stop is git-ignored and travels with gpurun). Run a tool or bench.py against it with CFEAR_HIP_LIB=tools/_stop/libcfear_hip_NAME.so.
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/cfear_radarodometry_code_public_amd/csrc
NAME=$1; EXTRA=$2
python -m cfear_radarodometry_code_public_amd.build > /dev/null
mkdir -p $R/tools/_stop
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable"
hipcc $FLAGS $EXTRA -c $C/pipeline.hip -o $R/tools/_stop/pipeline_$NAME.o &
hipcc $FLAGS $EXTRA -c $C/replay.hip -o $R/tools/_stop/replay_$NAME.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_stop/libcfear_hip_$NAME.so $R/tools/_stop/pipeline_$NAME.o $R/tools/_stop/replay_$NAME.o $(ls $C/*.o | grep -v -e pipeline.o -e replay.o)
rm -f $R/tools/_stop/pipeline_$NAME.o $R/tools/_stop/replay_$NAME.o
ls -la $R/tools/_stop/libcfear_hip_$NAME.so
