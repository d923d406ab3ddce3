# many resident sequences replaying a few sweep streams: which replicas deviate from the oracle, and is it repeatable run to
# run? DEFAULTSTREAM=1 shows what happens when torch produces the input on its default stream (handle 0) and that handle
# is passed to the library: the context creates a stream of its own, unordered with torch's.
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cfear_radarodometry_code_public_amd import capi, synth
from oracle import binding as ob
T, U, B = int(os.environ.get("T", "6")), 4, int(os.environ.get("B", "64"))
RR = np.float32(0.0595238)
kw = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, submap_scan_size=4, compensate=1, radar_ccw=0)
streams = [synth.world_sequence(T, seed=40 + u, world_seed=1300 + u, t0=11 * u)[0] for u in range(U)]
d_unique = torch.from_numpy(np.stack(streams)).cuda()
torch.cuda.synchronize()
idx = (torch.randperm(B, generator=torch.Generator().manual_seed(7)) % U).cuda()
kinds = idx.cpu().numpy()
if not os.environ.get('DEFAULTSTREAM'):
    torch.cuda.set_stream(torch.cuda.Stream())  # an explicit (non-default) stream shared by torch and the library
ctx = capi.Context(capi.default_params(**kw), 400, 3360, stream=torch.cuda.current_stream().cuda_stream)
print('stream handle', torch.cuda.current_stream().cuda_stream)
odo = ctx.odometry(B)
fus = [ob.Fuser(ob.default_params(**kw)) for u in range(U)]
exp_nc = np.zeros((T, U), int); exp_pose = np.zeros((T, U, 3))
for t in range(T):
    for u in range(U):
        exp_pose[t, u] = fus[u].process_polar(streams[u][t]); exp_nc[t, u] = len(fus[u].last_cells())
runs = []
for run in range(2):
    odo.reset()
    ncs = np.zeros((T, B), int); poses = np.zeros((T, B, 3))
    for t in range(T):
        d = d_unique[idx, t].contiguous()
        if os.environ.get('PRESYNC'): torch.cuda.synchronize()
        if os.environ.get('HOSTIN'): odo.step_host(d.cpu().numpy())
        else: odo.step_device(d.data_ptr())
        torch.cuda.synchronize()
        poses[t] = odo.poses(); ncs[t] = [odo.summary(q)[1] for q in range(B)]
    runs.append((ncs, poses))
    for t in range(T):
        badc = [(q, int(ncs[t, q]), int(exp_nc[t, kinds[q]])) for q in range(B) if ncs[t, q] != exp_nc[t, kinds[q]]]
        badp = [q for q in range(B) if np.abs(poses[t, q] - exp_pose[t, kinds[q]]).max() > 1e-4]
        print("run %d step %d: n_cells != oracle for %d sequences %s; pose off for %d" % (run, t, len(badc), badc[:5], len(badp)))
print("runs identical:", np.array_equal(runs[0][0], runs[1][0]), np.array_equal(runs[0][1], runs[1][1]))
