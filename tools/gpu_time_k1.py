# K1 (k-strongest) timing / tuning on the GPU box: interleaved A/B over (occupancy variant, rows per wave) settings
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfear_radarodometry_code_public_amd import capi, synth
A, R, k = 400, 3360, 12
n = int(os.environ.get("K1_N", "256"))
p = capi.default_params(range_res=np.float32(0.0595238))
ctx = capi.Context(p, A, R)
data = {}
data["uniform"] = torch.randint(0, 256, (n, A, R), dtype=torch.uint8, device="cuda")
w = synth.World(1234)
base = torch.from_numpy(np.stack([synth.world_scan(w, t, seed=1) for t in range(8)])).cuda()
data["world"] = base.repeat(n // 8, 1, 1).contiguous()
if os.environ.get("K1_TIES"):  # S-ties: five intensity levels, > 64 candidates tie at the threshold (the positional scan decides)
    tb = torch.from_numpy(np.stack([synth.ties_scan(A, R, seed=7 + u) for u in range(8)])).cuda()
    data["ties"] = tb.repeat(n // 8, 1, 1).contiguous()
if os.environ.get("K1_UNIFORM_SEEDED"):  # S-uniform as SURVEY 8(d) seeds it (0xC0FFEE + seq) instead of torch.randint
    ub = torch.from_numpy(np.stack([synth.uniform_scan(A, R, seed=0xC0FFEE + u) for u in range(min(n, 64))])).cuda()
    data["uniform"] = ub.repeat((n + ub.shape[0] - 1) // ub.shape[0], 1, 1)[:n].contiguous()
out = torch.zeros((n, A, k), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
configs = [tuple(int(x) for x in c.split(",")) for c in os.environ.get("K1_CONFIGS", "7,4").split(";")]
res = {}
for rep in range(int(os.environ.get("K1_REPS", "5"))):
    for (occ, over) in configs:
        ctx.tune(capi.TUNE_FILTER_OCCUPANCY, occ); ctx.tune(capi.TUNE_FILTER_ROWS_PER_WAVE, over)
        for name, d in data.items():
            t = ctx.time_kstrongest(d, n, out, 2, 10)
            res.setdefault((occ, over, name), []).append(t)
for key, ts in res.items():
    t = min(ts)
    print("occ=%d rows/wave=%d %-8s min %.1f us (med %.1f)  %.2f TB/s  %.0f scans/s" % (
        key[0], key[1], key[2], t * 1e6, np.median(ts) * 1e6, n * (A * R + A * k * 4) / t / 1e12, n / t))
