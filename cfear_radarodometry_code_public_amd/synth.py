"""Synthetic Navtech-style polar sweeps. Data generator for tests/bench only.

Three families, all uint8 [A, R] with rows = azimuth, cols = range bin (radar_driver.cpp:92-98):
  * uniform : iid uniform[0,255]  (bandwidth stress: ~76 % of bins >= z_min)
  * ties    : intensities quantised to a few levels (forces the range tie-break of
              radar_filters.cpp:224-228)
  * world   : 2-D polygonal world ray-cast from a sensor on a circular trajectory
              (1.0 m / 0.02 rad per frame), Gaussian range blobs over a noise floor.
"""
import os

import numpy as np


def uniform_scan(A=400, R=3360, seed=0xC0FFEE):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(A, R), dtype=np.uint8)


def ties_scan(A=400, R=3360, seed=7, levels=(0, 59, 60, 120, 255), p=None):
    rng = np.random.Generator(np.random.PCG64(seed))
    lv = np.asarray(levels, dtype=np.uint8)
    return lv[rng.choice(len(lv), size=(A, R), p=p)]


class World:
    """Outer 160 x 120 m rectangle + n random axis-unaligned boxes; segments as [S, 2, 2]."""

    def __init__(self, seed=1234, n_boxes=40, path_radius=50.0):
        rng = np.random.Generator(np.random.PCG64(seed))
        segs = []
        W, H = 80.0, 60.0
        c = np.array([[-W, -H], [W, -H], [W, H], [-W, H]])
        for i in range(4):
            segs.append([c[i], c[(i + 1) % 4]])
        n = 0
        while n < n_boxes:
            ctr = np.array([rng.uniform(-W + 8, W - 8), rng.uniform(-H + 8, H - 8)])
            hw, hh = rng.uniform(2.0, 7.5, size=2)
            if abs(np.hypot(*ctr) - path_radius) < 6.0 + np.hypot(hw, hh):
                continue  # keep the sensor path free
            ang = rng.uniform(0, np.pi)
            Rm = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
            cs = (Rm @ np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]]).T).T + ctr
            for i in range(4):
                segs.append([cs[i], cs[(i + 1) % 4]])
            n += 1
        self.segs = np.asarray(segs, dtype=np.float64)
        self.path_radius = path_radius


def gt_pose(t, step=1.0, yaw_rate=0.02, radius=None):
    """Pose (x, y, psi) of frame t (mid-sweep) on the circle of radius step/yaw_rate."""
    r = step / yaw_rate if radius is None else radius
    psi = yaw_rate * t
    return np.array([r * np.sin(psi), -r * np.cos(psi), psi])


def world_scan(world, t, A=400, R=3360, range_res=np.float32(0.0595238), seed=0, ccw=False,
               step=1.0, yaw_rate=0.02, z_min=60, distort=True):
    """One sweep at frame index t. Returns uint8 [A, R]."""
    rng = np.random.Generator(np.random.PCG64([seed, t]))
    res = float(np.float32(range_res))
    x0, y0, psi = gt_pose(t, step, yaw_rate)
    a = np.arange(A)
    theta = (a + 1) / A * 2 * np.pi  # radar_filters.cpp:317
    s = ((a + 1) / A - 0.5) if distort else np.zeros(A)
    if ccw:
        s = -s
    # sensor pose during the sweep = mid pose (+) s * (per-frame motion in the sensor frame)
    mx, my, mth = step, 0.0, yaw_rate
    ox = x0 + np.cos(psi) * (s * mx) - np.sin(psi) * (s * my)
    oy = y0 + np.sin(psi) * (s * mx) + np.cos(psi) * (s * my)
    ang = psi + s * mth + theta
    dx, dy = np.cos(ang), np.sin(ang)
    p0 = world.segs[:, 0, :]
    e = world.segs[:, 1, :] - p0
    # solve o + t d = p0 + u e
    den = dx[:, None] * e[None, :, 1] - dy[:, None] * e[None, :, 0]
    wx = p0[None, :, 0] - ox[:, None]
    wy = p0[None, :, 1] - oy[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        tt = (wx * e[None, :, 1] - wy * e[None, :, 0]) / den
        uu = (wx * dy[:, None] - wy * dx[:, None]) / den
    hit = (np.abs(den) > 1e-12) & (tt > 0.5) & (uu >= 0) & (uu <= 1)
    tt = np.where(hit, tt, np.inf)
    order = np.sort(tt, axis=1)[:, :2]
    img = np.clip(rng.normal(25.0, 8.0, size=(A, R)), 0, 255)
    # 0.1 % salt speckle >= z_min
    nsp = int(0.001 * A * R)
    img[rng.integers(0, A, nsp), rng.integers(0, R, nsp)] = rng.uniform(z_min, z_min + 60, nsp)
    bins = np.arange(R)
    power = rng.uniform(90, 200, size=A)
    for hno in range(2):
        rr = order[:, hno]
        ok = np.isfinite(rr)
        if hno == 1:
            ok &= rng.random(A) < 0.5
        b0 = (rr - res / 2) / res
        amp = power * (1.0 if hno == 0 else 0.45)
        for i in np.nonzero(ok)[0]:
            lo, hi = int(max(0, b0[i] - 8)), int(min(R, b0[i] + 9))
            if lo >= hi:
                continue
            blob = amp[i] * np.exp(-0.5 * ((bins[lo:hi] - b0[i]) / 2.0) ** 2)
            img[i, lo:hi] = np.maximum(img[i, lo:hi], blob)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def world_sequence(T, A=400, R=3360, range_res=np.float32(0.0595238), seed=0, world_seed=1234, ccw=False,
                   t0=0, **kw):
    """uint8 [T, A, R] sweeps and ground-truth poses [T, 3] relative to frame t0."""
    w = World(world_seed)
    imgs = np.stack([world_scan(w, t0 + t, A, R, range_res, seed, ccw, **kw) for t in range(T)])
    g0 = gt_pose(t0)
    c, s = np.cos(g0[2]), np.sin(g0[2])
    gts = []
    for t in range(T):
        g = gt_pose(t0 + t)
        d = g[:2] - g0[:2]
        gts.append([c * d[0] + s * d[1], -s * d[0] + c * d[1], g[2] - g0[2]])
    return imgs, np.asarray(gts)


def _stream_file(cache_dir, seed, T, A, R, range_res, ccw, t0=None):
    tag = "" if t0 is None else "_t%d" % t0
    return "%s/cfear_world_s%d%s_T%d_%dx%d_r%08x_c%d.npy" % (cache_dir, seed, tag, T, A, R, np.float32(range_res).view(np.uint32), int(ccw))


def _stream_worker(job):
    path, seed, T, A, R, range_res, ccw, t0, world_seed = job
    imgs, _ = world_sequence(T, A, R, np.float32(range_res), seed=seed, world_seed=world_seed, ccw=bool(ccw), t0=t0)
    tmp = "%s.%d.tmp.npy" % (path, os.getpid())
    np.save(tmp, imgs)
    os.replace(tmp, path)  # atomic: several ranks may want the same stream
    return path


def _run_jobs(jobs, procs):
    """jobs of _stream_worker in fresh interpreters (`python -m ...synth --gen`), at most `procs` at a time"""
    import subprocess
    import sys
    n = max(1, min(len(jobs), procs or len(os.sched_getaffinity(0))))
    if len(jobs) == 1 or n == 1:
        for j in jobs:
            _stream_worker(j)
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    running, todo = [], list(jobs)
    while todo or running:
        while todo and len(running) < n:
            j = todo.pop()
            running.append(subprocess.Popen([sys.executable, "-m", "cfear_radarodometry_code_public_amd.synth", "--gen"] + [str(x) for x in j],
                                            cwd=root, env=env))
        p = running.pop(0)
        if p.wait(timeout=3600) != 0:
            for q in running:
                q.kill()
            raise RuntimeError("synthetic stream worker failed")


def world_streams(seeds, T, A=400, R=3360, range_res=np.float32(0.0595238), ccw=True, procs=None, cache_dir=None):
    """uint8 [len(seeds), T, A, R]: one world sequence per seed (own world, own start point on the path), generated by
    worker processes (one sweep costs ~50 ms of one core) and cached as .npy files so that the next run maps them back in."""
    cache_dir = cache_dir or os.environ.get("CFEAR_SYNTH_CACHE", "/tmp/cfear_synth_cache")
    os.makedirs(cache_dir, exist_ok=True)
    files = [_stream_file(cache_dir, s, T, A, R, range_res, ccw) for s in seeds]
    jobs = [(f, s, T, A, R, float(range_res), int(ccw), 17 * (s % 64), 1234 + s) for f, s in zip(files, seeds) if not os.path.exists(f)]
    _run_jobs(jobs, procs)
    out = np.empty((len(seeds), T, A, R), dtype=np.uint8)
    for i, f in enumerate(files):
        out[i] = np.load(f, mmap_mode="r")
    return out


def world_sequence_long(T, A=400, R=3360, range_res=np.float32(0.0595238), seed=0, world_seed=1234, ccw=False, t0=0, chunk=25,
                        procs=None, cache_dir=None):
    """world_sequence for long runs: the same sweeps and ground truth, generated in chunks of frames by worker processes.
    Returns (list of read-only [<=chunk, A, R] arrays, ground-truth poses [T, 3])."""
    cache_dir = cache_dir or os.environ.get("CFEAR_SYNTH_CACHE", "/tmp/cfear_synth_cache")
    os.makedirs(cache_dir, exist_ok=True)
    starts = list(range(0, T, chunk))
    files = [_stream_file(cache_dir, seed * 100003 + world_seed, min(chunk, T - a), A, R, range_res, ccw, t0=t0 + a) for a in starts]
    jobs = [(f, seed, min(chunk, T - a), A, R, float(range_res), int(ccw), t0 + a, world_seed) for f, a in zip(files, starts) if not os.path.exists(f)]
    _run_jobs(jobs, procs)
    g0 = gt_pose(t0)
    c, s = np.cos(g0[2]), np.sin(g0[2])
    gts = []
    for t in range(T):
        g = gt_pose(t0 + t)
        d = g[:2] - g0[:2]
        gts.append([c * d[0] + s * d[1], -s * d[0] + c * d[1], g[2] - g0[2]])
    return [np.load(f, mmap_mode="r") for f in files], np.asarray(gts)


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 11 and sys.argv[1] == "--gen":
        a = sys.argv[2:]
        _stream_worker((a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]), int(a[6]), int(a[7]), int(a[8])))
