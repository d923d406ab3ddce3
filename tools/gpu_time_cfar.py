# the batched CA-CFAR detector alone: launches of B scans (400 x 3360), the preset of params/kstrong_vs_cfar/oxford-cfear-3-ca-cfar
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfear_radarodometry_code_public_amd import capi, synth
A, R, RR = 400, 3360, np.float32(0.0595238)
B = int(os.environ.get("CFAR_B", "1536")); U = 16
w = synth.World(1234)
uniq = torch.from_numpy(np.stack([synth.world_scan(w, u, A, R, RR, seed=1) for u in range(U)])).cuda()
d = uniq[torch.arange(B, device="cuda") % U].contiguous()
PRESETS = ((20.0, 40, 10, 0.01), (60.0, 10, 20, 0.01), (20.0, 500, 10, 0.0001))
if os.environ.get("CFAR_PRESET"): PRESETS = (PRESETS[int(os.environ["CFAR_PRESET"])],)  # (counters of one preset: tools/pmc_cfar.sh)
for zmin, win, guard, pfa in PRESETS:
    ctx = capi.Context(capi.default_params(range_res=RR, z_min=zmin), A, R, stream=torch.cuda.current_stream().cuda_stream)
    cap = 16384
    xyi = torch.empty((B, cap, 3), dtype=torch.float32, device="cuda"); cnt = torch.empty((B,), dtype=torch.int32, device="cuda")
    for it in range(3):
        ctx.filter_cfar_batch(d.data_ptr(), B, xyi.data_ptr(), cap, cnt.data_ptr(), window_size=win, nb_guard_cells=guard, false_alarm_rate=pfa)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for it in range(n):
        ctx.filter_cfar_batch(d.data_ptr(), B, xyi.data_ptr(), cap, cnt.data_ptr(), window_size=win, nb_guard_cells=guard, false_alarm_rate=pfa)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print("z_min %g window %d guard %d pfa %g: %.1f us per %d scans = %.2f TB/s of image bytes, %.0f detections per scan" %
          (zmin, win, guard, pfa, dt * 1e6, B, B * A * R / dt / 1e12, cnt.float().mean().item()))
    ctx.close()
