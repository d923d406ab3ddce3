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


def _patch_hash(seg, patch, salt):
    """deterministic uniform [0, 1) per (segment, patch along it): the world's reflectivity texture"""
    x = np.sin(seg * 12.9898 + patch * 78.233 + salt * 37.719) * 43758.5453
    return x - np.floor(x)


def render_scan(segs, pose, motion, A, R, range_res, rng, ccw=False, z_min=60, distort=True, hits=2, p_extra=0.5, sigma=2.0,
                amp_extra=0.45, texture=None):
    """One sweep of a sensor at `pose` = (x, y, psi) (mid-sweep) that moves by `motion` = (mx, my, mth) per sweep in its own
    frame, over the wall segments `segs` [S, 2, 2]. The first `hits` intersections of every ray give Gaussian range blobs (the
    first always, the later ones with probability p_extra each and amp_extra of the amplitude). Returns uint8 [A, R].
    texture = (patch length in m, fraction of strong patches, jitter): the echo's amplitude is a property of the place on the surface that was hit -
    patches along every segment are either strong scatterers (amplitude 140-220: window frames, pillars, parked metal) or weak ones (45-65: plain
    wall, mostly below z_min) - instead of a random number per azimuth and sweep; what survives the filter then sits at world-fixed spots."""
    res = float(np.float32(range_res))
    x0, y0, psi = pose
    a = np.arange(A)
    theta = (a + 1) / A * 2 * np.pi  # radar_filters.cpp:317
    s = ((a + 1) / A - 0.5) if distort else np.zeros(A)
    if ccw:
        s = -s
    # sensor pose during the sweep = mid pose (+) s * (per-frame motion in the sensor frame)
    mx, my, mth = motion
    ox = x0 + np.cos(psi) * (s * mx) - np.sin(psi) * (s * my)
    oy = y0 + np.sin(psi) * (s * mx) + np.cos(psi) * (s * my)
    ang = psi + s * mth + theta
    dx, dy = np.cos(ang), np.sin(ang)
    p0 = segs[:, 0, :]
    e = segs[:, 1, :] - p0
    # solve o + t d = p0 + u e
    den = dx[:, None] * e[None, :, 1] - dy[:, None] * e[None, :, 0]
    wx = p0[None, :, 0] - ox[:, None]
    wy = p0[None, :, 1] - oy[:, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        tt = (wx * e[None, :, 1] - wy * e[None, :, 0]) / den
        uu = (wx * dy[:, None] - wy * dx[:, None]) / den
    hit = (np.abs(den) > 1e-12) & (tt > 0.5) & (uu >= 0) & (uu <= 1)
    tt = np.where(hit, tt, np.inf)
    if texture is not None:
        nh = min(hits, tt.shape[1])
        idx = np.argpartition(tt, nh - 1, axis=1)[:, :nh] if nh < tt.shape[1] else np.tile(np.arange(tt.shape[1]), (A, 1))
        sub = np.take_along_axis(tt, idx, axis=1)
        o2 = np.argsort(sub, axis=1)
        idx = np.take_along_axis(idx, o2, axis=1)
        order = np.take_along_axis(sub, o2, axis=1)
        along = np.take_along_axis(np.where(hit, uu, 0.0), idx, axis=1) * np.hypot(e[:, 0], e[:, 1])[idx]
        patch = np.floor(along / texture[0])
        h1, h2 = _patch_hash(idx.astype(np.float64), patch, 1.0), _patch_hash(idx.astype(np.float64), patch, 2.0)
        tex_amp = np.where(h1 < texture[1], 140.0 + 80.0 * h2, 45.0 + 20.0 * h2)
        if order.shape[1] < hits:
            order = np.concatenate([order, np.full((A, hits - order.shape[1]), np.inf)], axis=1)
            tex_amp = np.concatenate([tex_amp, np.zeros((A, hits - tex_amp.shape[1]))], axis=1)
    else:
        order = np.sort(tt, axis=1)[:, :hits]
        if order.shape[1] < hits:
            order = np.concatenate([order, np.full((A, hits - order.shape[1]), np.inf)], axis=1)
    img = np.clip(rng.normal(25.0, 8.0, size=(A, R)), 0, 255)
    # 0.1 % salt speckle >= z_min
    nsp = int(0.001 * A * R)
    img[rng.integers(0, A, nsp), rng.integers(0, R, nsp)] = rng.uniform(z_min, z_min + 60, nsp)
    bins = np.arange(R)
    power = rng.uniform(90, 200, size=A)
    half = int(np.ceil(4.0 * sigma))
    for hno in range(hits):
        rr = order[:, hno]
        ok = np.isfinite(rr)
        if hno >= 1:
            ok &= rng.random(A) < p_extra
        b0 = (rr - res / 2) / res
        amp = power * (1.0 if hno == 0 else amp_extra)
        if texture is not None:
            amp = tex_amp[:, hno] * (1.0 + texture[2] * (power - 145.0) / 55.0) * (1.0 if hno == 0 else amp_extra)
        for i in np.nonzero(ok)[0]:
            lo, hi = int(max(0, b0[i] - half)), int(min(R, b0[i] + half + 1))
            if lo >= hi:
                continue
            blob = amp[i] * np.exp(-0.5 * ((bins[lo:hi] - b0[i]) / sigma) ** 2)
            img[i, lo:hi] = np.maximum(img[i, lo:hi], blob)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def world_scan(world, t, A=400, R=3360, range_res=np.float32(0.0595238), seed=0, ccw=False,
               step=1.0, yaw_rate=0.02, z_min=60, distort=True):
    """One sweep at frame index t of the circular trajectory. Returns uint8 [A, R]."""
    rng = np.random.Generator(np.random.PCG64([seed, t]))
    return render_scan(world.segs, gt_pose(t, step, yaw_rate), (step, 0.0, yaw_rate), A, R, range_res, rng, ccw=ccw, z_min=z_min,
                       distort=distort)


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


# ---- driving-like recordings (BASELINE configs[4]: a car in a city, not a constant-speed circle) -------------------------
# A closed track (straights + corner arcs of different radii) through one of three world families, driven with a speed
# schedule: stops of 20-40 sweeps (no new keyframe, zero-motion compensation), crawling, ramps between 0.2 and 3.5 m/sweep
# (keyframe every sweep), corners at up to +-0.15 rad/sweep, and reversing. odometrykeyframefuser.cpp:62-94,227-249 (keyframe
# rule, velocity sanity check, ring turnover) sees every regime.
class Track:
    """Closed centre line: rectangle with half extents (hx, hy) and one arc radius per corner, counter-clockwise from the
    middle of the bottom side. pose(s) -> (x, y, heading, curvature) at arc length s (periodic)."""

    def __init__(self, hx, hy, radii):
        r = [float(v) for v in radii]  # corners: bottom-right, top-right, top-left, bottom-left
        self.parts = []  # (kind, length, x, y, heading, curvature) at the start of the part
        c = [(hx, -hy), (hx, hy), (-hx, hy), (-hx, -hy)]
        hd = [0.0, np.pi / 2, np.pi, 3 * np.pi / 2]
        side = [2 * hx, 2 * hy, 2 * hx, 2 * hy]
        x, y = 0.0, -hy
        for i in range(4):
            rp, rn = r[(i - 1) % 4], r[i]
            ln = side[i] - rp - rn if i else side[0] / 2 - rn
            self.parts.append(("line", ln, x, y, hd[i], 0.0))
            x += ln * np.cos(hd[i]); y += ln * np.sin(hd[i])
            self.parts.append(("arc", rn * np.pi / 2, x, y, hd[i], 1.0 / rn))
            cx, cy = x - rn * np.sin(hd[i]), y + rn * np.cos(hd[i])
            h2 = hd[i] + np.pi / 2
            x, y = cx + rn * np.sin(h2), cy - rn * np.cos(h2)
        self.parts.append(("line", side[0] / 2 - r[3], x, y, 0.0, 0.0))
        self.length = sum(q[1] for q in self.parts)

    def pose(self, s):
        s = s % self.length
        for kind, ln, x, y, h, k in self.parts:
            if s <= ln:
                if kind == "line":
                    return x + s * np.cos(h), y + s * np.sin(h), h, 0.0
                r = 1.0 / k
                cx, cy = x - r * np.sin(h), y + r * np.cos(h)
                h2 = h + s * k
                return cx + r * np.sin(h2), cy - r * np.cos(h2), h2, k
            s -= ln
        return self.parts[0][2], self.parts[0][3], 0.0, 0.0

    def polyline(self, step=1.0):
        n = int(self.length / step) + 1
        return np.array([self.pose(i * self.length / n)[:2] for i in range(n)])


def _box(ctr, hw, hh, ang):
    Rm = np.array([[np.cos(ang), -np.sin(ang)], [np.sin(ang), np.cos(ang)]])
    cs = (Rm @ np.array([[-hw, -hh], [hw, -hh], [hw, hh], [-hw, hh]]).T).T + np.asarray(ctr)
    return [[cs[i], cs[(i + 1) % 4]] for i in range(4)]


DRIVE_KINDS = ("blocks", "canyon", "field", "thicket", "street")


class DriveWorld:
    """kind 'blocks': yard with an outer wall and scattered buildings (the family of World, ~200 surface points per sweep);
    'canyon': streets lined with facades (broken by gaps and alcoves), parked boxes and buildings behind them, several
    echoes per azimuth (>= 500 surface points per sweep); 'field': open ground, a few small objects and a far fence
    'thicket': the canyon's streets through a forest of 1500 small objects, ten equally strong echoes per azimuth - with k = 40 and
    a small `res` more than a thousand surface points per sweep (the reference's resolution sweep, params/resolution/oxford_cfear-3:16);
    (marginal registrations). render: keyword arguments of render_scan for this family."""

    def __init__(self, kind="blocks", seed=0):
        assert kind in DRIVE_KINDS
        rng = np.random.Generator(np.random.PCG64([77, seed, DRIVE_KINDS.index(kind)]))
        self.kind = kind
        segs = []
        if kind == "blocks":
            self.track = Track(48.0, 32.0, (10.0, 14.0, 22.0, 10.0))
            W, H = 80.0, 60.0
            segs += _box((0, 0), W, H, 0.0)
            self.render = dict(hits=2, p_extra=0.5, sigma=2.0)
            n_obj, size, clear = 46, (2.0, 7.5), 6.0
            lim = (W - 8, H - 8)
        elif kind in ("canyon", "thicket", "street"):
            self.track = Track(60.0, 40.0, (10.0, 12.0, 10.0, 16.0))
            self.render = dict(hits=5, p_extra=0.85, sigma=1.0, amp_extra=0.7)
            if kind == "street":  # the canyon with a reflectivity that belongs to the surfaces (round 6): what the reference's P2P preset needs
                # (patches of 1.5 m, a fifth of them strong: of the variants tried - profiles/r06_world_realism.json - the one on which P2P tracks best;
                # denser strong patches bring the grid-locking of the plain canyon back)
                self.render = dict(hits=5, p_extra=0.85, sigma=1.0, amp_extra=0.85, texture=(1.5, 0.2, 0.05))
            n_obj, size, clear = 320, (1.5, 6.0), 7.0
            lim = (150.0, 130.0)
            if kind == "thicket":  # the canyon's streets in a forest of small objects, every echo as strong as the first
                self.render = dict(hits=10, p_extra=1.0, sigma=1.5, amp_extra=0.95)
                n_obj, size, clear = 1500, (0.4, 1.6), 5.0
        else:
            self.track = Track(70.0, 45.0, (12.0, 25.0, 40.0, 10.0))
            W, H = 150.0, 120.0
            for _ in range(4):  # a few hedges instead of an outer wall
                c0 = np.array([rng.uniform(-W, W), rng.uniform(-H, H)])
                a0 = rng.uniform(0, np.pi)
                ln = rng.uniform(30.0, 70.0)
                segs.append([c0, c0 + ln * np.array([np.cos(a0), np.sin(a0)])])
            self.render = dict(hits=2, p_extra=0.3, sigma=2.0)
            n_obj, size, clear = 30, (0.6, 2.5), 5.0
            lim = (W - 10, H - 10)
        path = self.track.polyline(1.0)
        if kind in ("canyon", "thicket", "street"):  # facades 9-12 m left and right of the centre line, in pieces of 6-25 m with gaps and set-backs
            for side in (-1.0, 1.0):
                s = 0.0
                while s < self.track.length:
                    ln = rng.uniform(6.0, 25.0)
                    off = side * rng.uniform(9.0, 12.0)
                    pts = []
                    for u in np.arange(s, min(s + ln, self.track.length), 2.0):
                        x, y, h, _ = self.track.pose(u)
                        pts.append([x - off * np.sin(h), y + off * np.cos(h)])
                    for a, b in zip(pts[:-1], pts[1:]):
                        segs.append([np.array(a), np.array(b)])
                    s += ln + rng.uniform(2.0, 9.0)
        n = 0
        tries = 0
        while n < n_obj and tries < 20000:
            tries += 1
            ctr = np.array([rng.uniform(-lim[0], lim[0]), rng.uniform(-lim[1], lim[1])])
            hw, hh = rng.uniform(size[0], size[1], size=2)
            if np.min(np.hypot(path[:, 0] - ctr[0], path[:, 1] - ctr[1])) < clear + np.hypot(hw, hh):
                continue  # keep the road free
            segs += _box(ctr, hw, hh, rng.uniform(0, np.pi))
            n += 1
        self.segs = np.asarray(segs, dtype=np.float64)


def drive_plan(T, world, seed=0, v_max=3.5, yaw_max=0.15, acc=0.3):
    """Speed schedule along world.track -> (poses [T, 3] mid-sweep in the world frame, motions [T, 3] per sweep in the sensor
    frame, ground truth [T, 3] relative to sweep 0). The schedule is a seeded sequence of legs: cruise at 0.2 ... v_max
    m/sweep, stop for 20-40 sweeps, reverse; the speed follows it with |dv| <= acc per sweep and is capped so that the yaw
    rate in the corners stays within yaw_max rad/sweep."""
    rng = np.random.Generator(np.random.PCG64([99, seed]))
    tr = world.track
    legs = [("cruise", 40, 1.0), ("stop", 25, 0.0), ("cruise", 60, v_max), ("cruise", 50, 0.2), ("reverse", 35, -0.8), ("stop", 20, 0.0)]
    target = []
    while len(target) < T:
        if not legs:
            kind = rng.choice(["cruise", "cruise", "cruise", "cruise", "stop", "reverse"])
            if kind == "cruise":
                legs.append((kind, int(rng.integers(25, 120)), float(rng.choice([0.2, 0.6, 1.0, 1.0, 1.8, 2.6, v_max]))))
            elif kind == "stop":
                legs.append((kind, int(rng.integers(20, 41)), 0.0))
            else:
                legs += [("stop", 8, 0.0), (kind, int(rng.integers(20, 41)), -float(rng.uniform(0.4, 1.2))), ("stop", 8, 0.0)]
        kind, n, v = legs.pop(0)
        target += [v] * n
    s, v = 0.0, 0.0
    poses, motions = np.zeros((T, 3)), np.zeros((T, 3))
    for t in range(T):
        # speed limit of the curvature ahead (a car brakes before the corner): the tightest arc within the braking distance
        look = abs(v) * abs(v) / (2 * acc) + 2.0
        kmax = max(abs(tr.pose(s + np.sign(v if v else 1.0) * d)[3]) for d in np.linspace(0.0, look, 8))
        cap = yaw_max / kmax if kmax > 0 else v_max
        want = float(np.clip(target[t], -cap, cap))
        v += float(np.clip(want - v, -acc, acc))
        if abs(v) < 1e-9:
            v = 0.0
        s_mid = s + 0.5 * v
        x, y, h, k = tr.pose(s_mid)
        poses[t] = (x, y, h)
        motions[t] = (v, 0.0, v * k)
        s += v
    c, sn = np.cos(poses[0, 2]), np.sin(poses[0, 2])
    d = poses[:, :2] - poses[0, :2]
    gt = np.stack([c * d[:, 0] + sn * d[:, 1], -sn * d[:, 0] + c * d[:, 1], np.unwrap(poses[:, 2]) - poses[0, 2]], axis=1)
    return poses, motions, gt


def drive_scan(world, poses, motions, t, A=400, R=3768, range_res=np.float32(0.0438), seed=0, ccw=False, z_min=60):
    rng = np.random.Generator(np.random.PCG64([seed, 4242, t]))
    return render_scan(world.segs, poses[t], motions[t], A, R, range_res, rng, ccw=ccw, z_min=z_min, **world.render)


def _drive_chunk(job):
    kind, wseed, seed, T, t0, n, A, R, range_res, ccw = job
    w = DriveWorld(kind, wseed)
    poses, motions, _ = drive_plan(T, w, seed)
    return t0, np.stack([drive_scan(w, poses, motions, t, A, R, np.float32(range_res), seed, bool(ccw)) for t in range(t0, t0 + n)])


def drive_chunks(T, kind="blocks", world_seed=0, seed=0, A=400, R=3768, range_res=np.float32(0.0438), ccw=False, chunk=25, procs=None,
                 ahead=None):
    """Generator over (t0, uint8 [n, A, R]) chunks of a T-sweep drive, rendered by worker processes a few chunks ahead of
    the consumer (a sweep costs ~50 ms of one core; nothing is kept: 10 000 sweeps are 15 GB)."""
    import concurrent.futures as cf
    import multiprocessing as mp
    procs = procs or len(os.sched_getaffinity(0))
    ahead = ahead or 2 * procs
    jobs = [(kind, world_seed, seed, T, a, min(chunk, T - a), A, R, float(range_res), int(ccw)) for a in range(0, T, chunk)]
    with cf.ProcessPoolExecutor(max_workers=procs, mp_context=mp.get_context("spawn")) as ex:
        pending, nxt = [], 0
        while nxt < len(jobs) or pending:
            while nxt < len(jobs) and len(pending) < ahead:
                pending.append(ex.submit(_drive_chunk, jobs[nxt])); nxt += 1
            yield pending.pop(0).result()


if __name__ == "__main__":
    import sys
    if len(sys.argv) == 11 and sys.argv[1] == "--gen":
        a = sys.argv[2:]
        _stream_worker((a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]), int(a[6]), int(a[7]), int(a[8])))
