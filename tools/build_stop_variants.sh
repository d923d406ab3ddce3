#!/bin/bash
# Profile builds of the library whose feature kernel returns after phase k (features_dev.h CFEAR_STOP_AT): the VALU / LDS
# instruction counters of successive variants difference into per-phase counts (tools/pmc_feat_phases.sh on the GPU box).
# Only pipeline.hip differs; the other objects are the product build's. Output: tools/_stop/libcfear_hip_stop<k>.so
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/cfear_radarodometry_code_public_amd/csrc
python -m cfear_radarodometry_code_public_amd.build > /dev/null
mkdir -p $R/tools/_stop
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable"
for k in ${STOPS:-1 2 3 4 5 6 7}; do
  ( hipcc $FLAGS -DCFEAR_FEATURES_STOP=$k -c $C/pipeline.hip -o $R/tools/_stop/pipeline_stop$k.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_stop/libcfear_hip_stop$k.so $R/tools/_stop/pipeline_stop$k.o $(ls $C/*.o | grep -v pipeline.o) &&
    rm $R/tools/_stop/pipeline_stop$k.o ) &
done
wait
ls -la $R/tools/_stop
