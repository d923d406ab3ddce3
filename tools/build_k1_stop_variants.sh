#!/bin/bash
# Profile builds of the library whose k-strongest kernel ends every row after phase n (kstrongest.hip CFEAR_K1_STOP): the VALU / SALU /
# LDS counters of successive variants difference into per-phase counts (tools/pmc_k1_phases.sh on the GPU box). Only kstrongest.hip
# differs; the other objects are the product build's. Output: tools/_stop/libcfear_hip_k1stop<n>.so
set -e
R=$(cd $(dirname $0)/.. && pwd); C=$R/cfear_radarodometry_code_public_amd/csrc
python -m cfear_radarodometry_code_public_amd.build > /dev/null
mkdir -p $R/tools/_stop
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable"
for n in ${STOPS:-1 2 3 4}; do
  ( hipcc $FLAGS -DCFEAR_K1_STOP=$n -c $C/kstrongest.hip -o $R/tools/_stop/kstrongest_stop$n.o &&
    hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_stop/libcfear_hip_k1stop$n.so $R/tools/_stop/kstrongest_stop$n.o $(ls $C/*.o | grep -v kstrongest.o) &&
    rm $R/tools/_stop/kstrongest_stop$n.o ) &
done
wait
ls -la $R/tools/_stop | grep k1stop
