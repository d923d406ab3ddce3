# experiment: aggregate throughput of S independent odometry batches on S HIP streams (do the filter, features and
# registration kernels of different batches overlap?)   MS_CONFIGS="streams x sequences;..."
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cfear_radarodometry_code_public_amd import capi
frames = 30
world = bench.make_streams(4, frames, 0)
d_unique = torch.from_numpy(world).cuda()
p = bench.params(capi)
for cfg in os.environ.get("MS_CONFIGS", "1x1024;2x512;4x256;1x768;2x768").split(";"):
    S, B = [int(v) for v in cfg.split("x")]
    idx = torch.arange(B, device="cuda") % 4
    d_polar = d_unique[idx].permute(1, 0, 2, 3).contiguous()  # [T][B][A][R]
    streams = [torch.cuda.Stream() for _ in range(S)]
    ctxs = [capi.Context(p, 400, 3360, stream=s.cuda_stream) for s in streams]
    odos = [c.odometry(B) for c in ctxs]
    torch.cuda.synchronize()
    def run(t0, t1):
        for t in range(t0, t1):
            for o in odos:
                o.step_device(d_polar[t].data_ptr())
    run(0, 10)
    torch.cuda.synchronize(); a = time.perf_counter()
    run(10, frames)
    torch.cuda.synchronize(); b = time.perf_counter()
    print("%d streams x %d sequences: %.0f scans/s  (%.3f ms per step of %d scans)" % (S, B, S * B * (frames - 10) / (b - a), (b - a) / (frames - 10) * 1e3, S * B), flush=True)
    del odos, ctxs
