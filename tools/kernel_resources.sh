#!/bin/bash
# VGPR / LDS / occupancy of every kernel (compiler remarks)
cd /tmp; for f in pipeline kstrongest; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c --cuda-device-only -Rpass-analysis=kernel-resource-usage $1/cfear_radarodometry_code_public_amd/csrc/$f.hip -I$1/include -o /tmp/kr_$f.o 2>&1 | grep -E "Function Name|VGPRs:|LDS Size|ScratchSize" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - ; done
