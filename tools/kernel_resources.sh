#!/bin/bash
# VGPR / AGPR / scratch / LDS / occupancy of every kernel (compiler remarks).  usage: kernel_resources.sh <repo root>
cd /tmp; for f in pipeline kstrongest cfar; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c --cuda-device-only -Rpass-analysis=kernel-resource-usage $1/cfear_radarodometry_code_public_amd/csrc/$f.hip -I$1/include -o /tmp/kr_$f.o 2>&1 | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' | paste - - - - - - ; done
