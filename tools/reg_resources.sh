#!/bin/bash
# Register / scratch / occupancy table of the batched registration kernels (register_step.hip) under the compile variants the round-4
# review asked about: the unified 512-entry register file of gfx950 is shared by VGPRs and AGPRs (occupancy is set by their SUM), so
# "spill to AGPRs" can only help below the occupancy-3 budget of 168. Output: what the code object says for every variant.
#   usage: tools/reg_resources.sh > profiles/r05_reg_resources.txt
R=$(cd $(dirname $0)/.. && pwd); C=$R/cfear_radarodometry_code_public_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Wno-unused-function -Wno-unused-variable -I$R/include"
row() {  # name, extra flags
  echo "== $1   ($2)"
  /opt/rocm/bin/hipcc $FLAGS $2 -c --cuda-device-only -Rpass-analysis=kernel-resource-usage $C/register_step.hip -o /tmp/rr.o 2>&1 |
    grep -E "Function Name|VGPRs:|AGPRs:|Spill|ScratchSize|Occupancy|LDS Size" | sed -e 's/.*remark: *//' -e 's/\[-Rpass.*//' -e 's/_ZN12_GLOBAL__N_120register_step_kernelI\(L[a-z0-9]*\)E\(L[a-z0-9]*\)E.*/kernel<\1,\2>/' | paste - - - - - - - - |
    grep -v "Lb1E"
  /opt/rocm/bin/hipcc $FLAGS $2 -S --cuda-device-only $C/register_step.hip -o /tmp/rr.s 2>/dev/null
  awk '/^_Z[A-Za-z0-9_]*:/{name=$1} /^; NumVgprs:/{v=$3} /^; NumAgprs:/{a=$3} /^; ScratchSize:/{print "   fn", substr(name,1,58), "vgpr", v, "agpr", a, "scratch", $3}' /tmp/rr.s |
    sed -e 's/_ZN9cfear_dev//' -e 's/_ZN12_GLOBAL__N_1//' | grep -E "associate_cell4|emit_blockILi1|emit_itemILi1|evaluate_partial_cILi1ELi1|ctl_after_candidate|ctl_after_it0"
  echo "   v_accvgpr_write / read in the object: $(grep -c v_accvgpr_write /tmp/rr.s) / $(grep -c v_accvgpr_read /tmp/rr.s);  scratch_store / scratch_load: $(grep -c scratch_store /tmp/rr.s) / $(grep -c scratch_load /tmp/rr.s)"
}
row "production: 256 threads, 3 workgroups per unit (168 registers)" ""
row "spills to AGPRs forced on" "-mllvm -amdgpu-spill-vgpr-to-agpr=1"
row "spills to AGPRs off" "-mllvm -amdgpu-spill-vgpr-to-agpr=0"
row "4 workgroups per unit (128 registers, 420 LDS matches)" "-DCFEAR_REG_MIN_WG=4 -DCFEAR_MATCH_LDS_CAP=420"
row "4 per unit + spills to AGPRs forced on" "-DCFEAR_REG_MIN_WG=4 -DCFEAR_MATCH_LDS_CAP=420 -mllvm -amdgpu-spill-vgpr-to-agpr=1"
row "2 per unit (256 registers: what the code wants unconstrained)" "-DCFEAR_REG_MIN_WG=2"
