# where the time of the registration's command loop goes (cfear_odometry_phase_times mode 4), median over the sequences
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.cuda.set_stream(torch.cuda.Stream())
from cfear_radarodometry_code_public_amd import capi
B = int(os.environ.get("ODO_B", "1536")); frames = int(os.environ.get("ODO_FRAMES", "16"))
SUBMAP = int(os.environ.get("ODO_SUBMAP", "0"))  # e.g. 50: the large-submap presets (and the 512-thread kernel below 257 sequences or from 24 keyframes)
U = int(os.environ.get('ODO_U', '16'))
streams = bench.make_streams(U, frames, 0)
d_unique = torch.from_numpy(streams).cuda()
idx = torch.arange(B, device="cuda") % U
P = bench.params(capi)
if SUBMAP: P.submap_scan_size = SUBMAP
ctx = capi.Context(P, 400, 3360, stream=torch.cuda.current_stream().cuda_stream)
odo = ctx.odometry(B)
odo.phase_times(None, controller=True)
for t in range(frames):
    odo.step_device(d_unique[idx, t].contiguous())
    torch.cuda.synchronize()
    buf = odo.phase_times(True)
    if t >= frames - 3:
        a = buf[:, :8].astype(np.float64) / 100.0  # us
        n = np.maximum(a[:, 3], 1)
        names = ["wait at command barrier", "command (evaluation) in wave 0", "wait at result barrier", "commands", "state function (gather + decide + step)",
                 "trust-region steps on their own", "publish (candidate / build / finish)", "ctl_lm_done"]
        print("frame %d: evaluation commands per registration: median %d" % (t, np.median(a[:, 3] * 100)))
        for i in (0, 1, 2, 4, 5, 6, 7):
            print("   %-40s total %.1f us   per command %.2f us" % (names[i], np.median(a[:, i]), np.median(a[:, i] / (a[:, 3] * 100).clip(1))))
        reg = buf[:, 14:29].astype(np.float64); tr = np.array([(r[r > 0][-1] - r[0]) / 100.0 for r in reg if (r > 0).sum() > 1])
        print("   registration per workgroup: median %.1f us" % np.median(tr))
