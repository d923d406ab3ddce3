# per-phase timing of odometry_step_kernel (wall_clock64 stamps by thread 0 of every block)
import ctypes as C, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.cuda.set_stream(torch.cuda.Stream())  # explicit stream shared with the library (handle 0 = torch's default stream would make the context create its own, unordered with torch)
from cfear_radarodometry_code_public_amd import capi
B = int(os.environ.get("ODO_B", "256")); frames = 16
U = int(os.environ.get('ODO_U', '4'))
streams = bench.dense_streams(U, frames) if os.environ.get('ODO_DENSE') else bench.make_streams(U, frames, 0)
d_unique = torch.from_numpy(streams).cuda()
idx = torch.arange(B, device="cuda") % U
p = bench.params(capi)
if os.environ.get('ODO_K'):
    p.k_strongest = int(os.environ['ODO_K'])
if os.environ.get('ODO_COST'):
    p.cost = int(os.environ['ODO_COST']); p.regularization = 1.0; p.covar_scale = 1.0
ctx = capi.Context(p, 400, 3360, stream=torch.cuda.current_stream().cuda_stream)
odo = ctx.odometry(B)
odo.phase_times(None)  # timed kernel instantiations from here on
for t in range(frames):
    d = d_unique[idx, t].contiguous()
    odo.step_device(d)
    torch.cuda.synchronize()
    buf = odo.phase_times(True)
    if t >= 12:
        ts = buf.astype(np.float64)
        acc = ts[:, 29:32].copy(); ts[:, 29:32] = 0
        n = (ts > 0).sum(1)
        q = int(np.argmax(n))
        S, nc, nk = odo.summary(q)
        feat = ts[q][:14]; feat = feat[feat > 0]      # features kernel: slots 0..13
        reg = ts[q][14:29]; reg = reg[reg > 0]         # registration kernel: slots 14..28
        fn = ["cloud", "compensate", "minmax+keys", "sort", "segments", "centroids", "cells-ranges", "cells-chunks", "cells-acc", "cells-epi", "compact", "grid"]
        rn = ["setup"] + [s + str(k) for k in range(1, 9) for s in ("build", "LM")]
        print("frame %d seq %d: cells %d kf %d outer %d inner %s  features %.1f us  registration %.1f us" %
              (t, q, nc, nk, S.outer_iterations, list(S.inner_iterations[:8]), (feat[-1] - feat[0]) / 100.0, (reg[-1] - reg[0]) / 100.0 if len(reg) > 1 else 0.0))
        print("   features:     " + "  ".join("%s %.1f" % (fn[i] if i < len(fn) else "f%d" % i, d) for i, d in enumerate(np.diff(feat) / 100.0)))
        print("   registration: " + "  ".join("%s %.1f" % (rn[i] if i < len(rn) else "r%d" % i, d) for i, d in enumerate(np.diff(reg) / 100.0)))
        print("   LM: evals %d  eval %.2f us each  controller %.2f us each" % (acc[q][2], acc[q][0] / 100.0 / max(acc[q][2], 1), acc[q][1] / 100.0 / max(acc[q][2], 1)))
        tf = np.array([(r[:14][r[:14] > 0][-1] - r[0]) / 100.0 for r in ts if (r[:14] > 0).sum() > 1])
        tr = np.array([(r[14:29][r[14:29] > 0][-1] - r[14]) / 100.0 for r in ts if (r[14:29] > 0).sum() > 1])
        print("   per workgroup, us: features min %.1f med %.1f max %.1f   registration min %.1f med %.1f max %.1f" %
              (tf.min(), np.median(tf), tf.max(), tr.min(), np.median(tr), tr.max()))
