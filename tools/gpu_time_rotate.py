# HBM throughput of the rotate kernel (range-major -> azimuth-major), 256 images per launch
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfear_radarodometry_code_public_amd import capi
torch.cuda.set_stream(torch.cuda.Stream())  # explicit stream shared with the library (handle 0 = torch's default stream would make the context create its own, unordered with torch)
n = 256
ctx = capi.Context(capi.default_params(), 400, 3360, stream=torch.cuda.current_stream().cuda_stream)
d_in = torch.randint(0, 256, (n, 3360, 400), dtype=torch.uint8, device="cuda")
d_out = torch.empty((n, 400, 3360), dtype=torch.uint8, device="cuda")
L = capi.lib()
for _ in range(3):
    L.cfear_rotate_polar_device(ctx.handle, d_in.data_ptr(), n, 3360, 400, d_out.data_ptr())
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    L.cfear_rotate_polar_device(ctx.handle, d_in.data_ptr(), n, 3360, 400, d_out.data_ptr())
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print("rotate: %.1f us per %d images, %.2f TB/s (read + write)" % (dt * 1e6, n, 2 * n * 3360 * 400 / dt / 1e12))
