"""One sequence replayed through cfear_odometry_replay_host (what bench.py's single_sequence.replay leg times), on its own so that a
rocprofv3 --kernel-trace --stats of this script shows the replay's kernels only: python tools/gpu_replay_rate.py [sweeps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cfear_radarodometry_code_public_amd import capi

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
imgs = bench.make_streams(1, 72, 0)[0]
T = imgs.shape[0]
ctx = capi.Context(bench.params(capi), bench.A, bench.R)
odo = ctx.odometry(1)
buf = ctx.pinned((n, 1, bench.A, bench.R))
period = 2 * (T - 1)
for i in range(n):
    m = i % period
    buf[i, 0] = imgs[m if m < T else period - m]
odo.replay_host(buf[:64])
odo.reset()
t0 = time.perf_counter()
rec = odo.replay_host(buf)
dt = time.perf_counter() - t0
print("replay: %d sweeps, %.0f sweeps/s, %.1f us per sweep; outer iterations mean %.2f, residuals mean %.0f" %
      (n, n / dt, 1e6 * dt / n, rec["outer_iterations"][1:].mean(), rec["num_residuals"][1:].mean()))
