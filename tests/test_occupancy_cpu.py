"""Occupancy guard: the step kernels are tuned to a number of resident wavefronts that a few extra registers or LDS bytes
silently take away (features: 128 VGPRs and 80,384 B of LDS for two 512-thread workgroups per CU; registration: 168 VGPRs incl. AGPRs and 53,760 B of LDS for three workgroups per CU; filter: 72 VGPRs for
seven waves per SIMD). Reads the compiler's kernel-resource remarks for gfx950 - no GPU needed."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cfear_radarodometry_code_public_amd", "csrc")


def remarks(src, tmp_path):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c", "--cuda-device-only",
           "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, src), "-I" + os.path.join(ROOT, "include"), "-o", str(tmp_path / "k.o")]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    kernels, cur = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = kernels.setdefault(m.group(1), {})
        m = re.search(r"remark:\s+(VGPRs|AGPRs|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|ScratchSize \[bytes/lane\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split(" ")[0]] = int(m.group(2))
    return kernels


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not installed")
def test_step_kernels_keep_their_occupancy(tmp_path):
    k = remarks("pipeline.hip", tmp_path)
    feat = [v for n, v in k.items() if "features_step_kernelILb0" in n][0]
    regs = {"wide:" + n: v for n, v in k.items() if "register_step64_kernelILb0" in n}  # the same shape compiled for large submaps (64 scans), one per cost metric
    regs.update({"small:" + n: v for n, v in remarks("register_step.hip", tmp_path).items() if "register_step_kernelILb0" in n})  # production (<= 8 scans): one per cost metric
    assert len(regs) == 6, sorted(regs)
    for n, reg in regs.items():
        assert reg["Occupancy"] >= 3 and reg["LDS"] <= 53760, (n, reg)      # three 256-thread workgroups per CU
    big = {n: v for n, v in remarks("register_step_large.hip", tmp_path).items() if "register_step_large_kernelILb0" in n}
    assert len(big) == 3, sorted(big)
    for n, reg in big.items():  # one 512-thread workgroup with the unit's LDS to itself: two waves per SIMD, nothing spilled but the frame of its callees
        assert reg["Occupancy"] >= 2 and reg["LDS"] <= 160 * 1024 and reg["ScratchSize"] <= 32, (n, reg)
    assert feat["Occupancy"] >= 4 and feat["LDS"] <= 80384 and feat["ScratchSize"] <= 16, feat  # two 512-thread workgroups per CU; at most two registers parked in scratch once per workgroup
    k = remarks("kstrongest.hip", tmp_path)
    flt = [v for n, v in k.items() if "kstrongest_kernelILi4ELi7" in n][0]
    assert flt["Occupancy"] >= 7 and flt["ScratchSize"] == 0, flt   # seven waves per SIMD, no spills
