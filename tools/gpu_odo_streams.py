# throughput of the batched odometry with the filter one sweep ahead on its own stream or not: ODO_CFG="overlap,sequences;..."
import ctypes as C, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
torch.cuda.set_stream(torch.cuda.Stream())  # explicit stream shared with the library (handle 0 = torch's default stream would make the context create its own, unordered with torch)
from cfear_radarodometry_code_public_amd import capi
frames = int(os.environ.get('ODO_FRAMES', '30'))
world = bench.make_streams(4, frames, 0)
d_unique = torch.from_numpy(world).cuda()
p = bench.params(capi)
for cfg in os.environ.get("ODO_CFG", "0,1536;1,1536;0,768;1,768").split(";"):
    S, B = [int(v) for v in cfg.split(",")]
    idx = torch.arange(B, device="cuda") % 4
    d_polar = d_unique[idx].permute(1, 0, 2, 3).contiguous()
    ctx = capi.Context(p, 400, 3360, stream=torch.cuda.current_stream().cuda_stream)
    odo = ctx.odometry(B, overlap=bool(S))
    for t in range(10):
        odo.step_device(d_polar[t].data_ptr())
    ctx.synchronize(); torch.cuda.synchronize(); a = time.perf_counter()
    for t in range(10, frames):
        odo.step_device(d_polar[t].data_ptr())
    ctx.synchronize(); torch.cuda.synchronize(); b = time.perf_counter()
    print("overlap %d sequences %d: %.0f scans/s (%.3f ms/step)" % (S, B, B * (frames - 10) / (b - a), (b - a) / (frames - 10) * 1e3), flush=True)
    odo.release(); del ctx, d_polar
    torch.cuda.empty_cache()
