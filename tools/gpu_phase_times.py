# per-phase timing of odometry_step_kernel (wall_clock64 stamps by thread 0 of every block)
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cfear_radarodometry_code_public_amd import capi
B = int(os.environ.get("ODO_B", "256")); frames = 16
streams = bench.make_streams(4, frames, 0)
d_unique = torch.from_numpy(streams).cuda()
idx = torch.arange(B, device="cuda") % 4
p = bench.params(capi)
ctx = capi.Context(p, 400, 3360, stream=torch.cuda.current_stream().cuda_stream)
odo = ctx.odometry(B)
L = capi.lib()
L.cfear_odometry_phase_times.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
L.cfear_odometry_phase_times(ctx.handle, odo._h, None)  # allocate
buf = np.zeros((B, 32), dtype=np.int64)
for t in range(frames):
    d = d_unique[idx, t].contiguous()
    odo.step_device(d)
    torch.cuda.synchronize()
    L.cfear_odometry_phase_times(ctx.handle, odo._h, buf.ctypes.data)
    if t >= 12:
        ts = buf.astype(np.float64)
        acc = ts[:, 29:32].copy(); ts[:, 29:32] = 0
        n = (ts > 0).sum(1)
        q = int(np.argmax(n))
        S, nc, nk = odo.summary(q)
        row = ts[q][:n[q]]
        d_us = np.diff(row) / 100.0
        names = ["cloud", "compensate", "minmax+keys", "sort", "segments", "centroids", "cells-ranges", "cells-chunks", "cells-acc", "cells-epi", "compact", "grid"]
        print("   nv(voxels) =", odo_nv if False else "")
        print("frame %d seq %d: cells %d kf %d outer %d inner %s total %.1f us" % (t, q, nc, nk, S.outer_iterations, list(S.inner_iterations[:8]), (row[-1] - row[0]) / 100.0))
        print("   " + "  ".join("%s %.1f" % (names[i] if i < len(names) else "r%d" % (i - len(names)), d_us[i]) for i in range(len(d_us))))
        print("   LM: evals %d  eval %.2f us each  controller %.2f us each" % (acc[q][2], acc[q][0] / 100.0 / max(acc[q][2], 1), acc[q][1] / 100.0 / max(acc[q][2], 1)))
        allt = np.array([(ts[b][:n[b]][-1] - ts[b][0]) / 100.0 for b in range(B) if n[b] > 1])
        print("   per-block total us: min %.1f med %.1f max %.1f" % (allt.min(), np.median(allt), allt.max()))
