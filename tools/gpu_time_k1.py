# scratch script (not committed): quick K1 timing on the GPU box
import time, numpy as np, torch, sys
sys.path.insert(0, '.')
from cfear_radarodometry_code_public_amd import capi, synth
A, R, k = 400, 3360, 12
p = capi.default_params(range_res=np.float32(0.0595238))
ctx = capi.Context(p, A, R)
for name, n in (("uniform", 256), ("world", 256)):
    if name == "uniform":
        d = torch.randint(0, 256, (n, A, R), dtype=torch.uint8, device="cuda")
    else:
        w = synth.World(1234)
        base = torch.from_numpy(np.stack([synth.world_scan(w, t, seed=1) for t in range(8)])).cuda()
        d = base.repeat(n // 8, 1, 1).contiguous()
    out = torch.zeros((n, A, k), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t = ctx.time_kstrongest(d, n, out, 3, 20)
    print(name, "n=%d  %.1f us/launch  %.2f TB/s  %.0f scans/s" % (n, t * 1e6, n * (A * R + A * k * 4) / t / 1e12, n / t))
