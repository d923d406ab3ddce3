# odometry throughput for k = 12 / 20 / 40 (k > 12: the general cloud pass, A * k > 5120 slots)
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.cuda.set_stream(torch.cuda.Stream())
from cfear_radarodometry_code_public_amd import capi
B, frames = int(os.environ.get('K_B', '1536')), 14
streams = bench.make_streams(4, frames, 0)
d_unique = torch.from_numpy(streams).cuda()
idx = torch.arange(B, device="cuda") % 4
for k in (1, 4, 12, 13, 20, 40, 64):
    p = bench.params(capi); p.k_strongest = k
    ctx = capi.Context(p, 400, 3360, stream=torch.cuda.current_stream().cuda_stream)
    odo = ctx.odometry(B)
    d = [d_unique[idx, t].contiguous() for t in range(frames)]
    for t in range(4): odo.step_device(d[t].data_ptr())
    ctx.synchronize(); torch.cuda.synchronize(); a = time.perf_counter()
    odo.profile(True)
    for t in range(4, frames): odo.step_device(d[t].data_ptr())
    ctx.synchronize(); torch.cuda.synchronize(); b = time.perf_counter()
    tf, nf = odo.profile_read(); tfe, tre, ns = odo.profile_read_stages()
    S, nc, nk = odo.summary(0)
    print("k=%d: %.0f scans/s  %.3f ms/step of %d  filter %.0f us features %.0f us registration %.0f us  cells %d" % (k, B * (frames - 4) / (b - a), (b - a) / (frames - 4) * 1e3, B, tf / nf * 1e6, tfe / ns * 1e6, tre / ns * 1e6, nc), flush=True)
    odo.release(); del ctx
