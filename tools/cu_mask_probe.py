"""What do the bits of a HIP compute-unit mask select on this GPU? Times the k-strongest filter alone (HBM-bound, scales with the
units it may use) on streams created with hipExtStreamCreateWithCUMask for a few bit patterns."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfear_radarodometry_code_public_amd import capi  # noqa: E402

hip = C.CDLL([m for m in open("/proc/self/maps").read().split() if "libamdhip64" in m][0])
A, R, B = 400, 3360, 1536
d_polar = torch.randint(0, 256, (B, A, R), dtype=torch.uint8, device="cuda")
d_slots = torch.empty((B, A, 12), dtype=torch.int32, device="cuda")
NCU = torch.cuda.get_device_properties(0).multi_processor_count


def run(name, bits):
    words = (NCU + 31) // 32
    mask = (C.c_uint32 * words)()
    for i in bits:
        mask[i // 32] |= 1 << (i % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), words, mask)
    assert rc == 0, rc
    ctx = capi.Context(capi.default_params(range_res=np.float32(0.0595238), z_min=60.0), A, R, stream=st.value)
    t = ctx.time_kstrongest(d_polar.data_ptr(), B, d_slots.data_ptr(), 2, 8)
    print("%-44s bits %3d: %7.1f us  %6.0f GB/s" % (name, len(bits), 1e6 * t, (A * R + A * 48) * B / t / 1e9), flush=True)
    ctx.close()
    hip.hipStreamDestroy(st)


run("all", list(range(NCU)))
for F in (32, 64, 128):
    run("low bits [0, F)", list(range(F)))
    run("every (NCU/F)-th bit", [i * (NCU // F) for i in range(F)])
    run("F/8 low bits of every 32-bit word", [w * 32 + b for w in range(8) for b in range(F // 8)])
    run("bits i with i % 8 < F/32 (whole residue classes)", [i for i in range(NCU) if i % 8 < F // 32])
run("bits 0..31 only", list(range(32)))
run("bits i % 8 == 0", [i for i in range(NCU) if i % 8 == 0])
run("bits i % 32 == 0", [i for i in range(NCU) if i % 32 == 0])
