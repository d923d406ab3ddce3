#!/bin/bash
# A/B builds of the library for measurements on one GPU box: tools/build_variant.sh NAME "-DMACRO=1 ..." compiles pipeline.hip,
# replay.hip, register_step.hip and register_step_large.hip with the extra flags and links them with the product build's other objects into tools/_stop/libcfear_hip_NAME.so
# (tools/_stop is git-ignored and travels with gpurun). Run a tool or bench.py against it with CFEAR_HIP_LIB=tools/_stop/libcfear_hip_NAME.so.
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/cfear_radarodometry_code_public_amd/csrc
NAME=$1; EXTRA=$2
python -m cfear_radarodometry_code_public_amd.build > /dev/null
mkdir -p $R/tools/_stop
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable"
TUS="pipeline replay register_step register_step_large"  # every translation unit that compiles the registration code
OBJS=""; SKIP=""
for t in $TUS; do
  hipcc $FLAGS $EXTRA -c $C/$t.hip -o $R/tools/_stop/${t}_$NAME.o &
  OBJS="$OBJS $R/tools/_stop/${t}_$NAME.o"; SKIP="$SKIP -e /$t.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_stop/libcfear_hip_$NAME.so $OBJS $(ls $C/*.o | grep -v $SKIP)
rm -f $OBJS
ls -la $R/tools/_stop/libcfear_hip_$NAME.so
