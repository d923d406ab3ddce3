"""Helper of tests/test_drive_replay_gpu.py and tests/run_drive_parity.py (not a test module): replays a driving-like synthetic
recording (synth.DriveWorld / drive_plan: stops, crawling, ramps to 3.5 m/sweep, corners at +-0.15 rad/sweep, reversing) through
cfear_odometry_replay_host on the device and through the oracle's fuser on the CPU, sweep by sweep, and lists every sweep at
which the two disagree. Test infrastructure: imports oracle/."""
import time

import numpy as np

from cfear_radarodometry_code_public_amd import capi, kitti, synth

A, R, RR = 400, 3768, np.float32(0.0438)  # the Oxford shape (BASELINE configs[4])
BASE = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, compensate=1, radar_ccw=0, cost=1, loss=1,
            loss_limit=0.1, submap_scan_size=4)


def regimes(motions):
    v, w = motions[:, 0], motions[:, 2]
    return {"stopped": v == 0, "crawl": (np.abs(v) > 0) & (np.abs(v) < 0.4), "reverse": v < 0, "fast": v > 3.0, "turn": np.abs(w) > 0.1}


def run(oracle, T, kind, world_seed=0, seed=1, piece=250, params=None, device=0, procs=None, log=None, persistent_max=None):
    """-> dict(mismatches=[(sweep, what, device, oracle)], poses_dev, poses_cpu, gt, cells, seconds...)"""
    kw = dict(BASE)
    kw.update(params or {})
    cfar = kw.pop("cfar", None)  # dict(window_size, nb_guard_cells, false_alarm_rate): CA-CFAR in front of the fuser instead of k-strongest
    world = synth.DriveWorld(kind, world_seed)
    poses_w, motions, gt = synth.drive_plan(T, world, seed)
    fu = oracle.Fuser(oracle.default_params(**kw))
    hip_kw = dict(kw)
    if cfar:
        hip_kw.update(filter_type=capi.FILTER_CACFAR, cfar_window_size=cfar["window_size"], cfar_nb_guard_cells=cfar["nb_guard_cells"],
                      cfar_false_alarm_rate=cfar["false_alarm_rate"])
    ctx = capi.Context(capi.default_params(**hip_kw), A, R, device=device)
    if persistent_max is not None:
        ctx.tune(capi.TUNE_REPLAY_PERSISTENT_MAX, persistent_max)
    odo = ctx.odometry(1)
    buf = ctx.pinned((piece, 1, A, R))
    mism, dev_poses, cpu_poses, cells, nkfs = [], [], [], [], []
    t_dev = t_cpu = 0.0
    fill, base = 0, 0
    t_start = time.time()

    def flush():
        nonlocal fill, base, t_dev, t_cpu
        if fill == 0:
            return
        t0 = time.time()
        rec = odo.replay_host(buf[:fill])[:, 0]
        t_dev += time.time() - t0
        t0 = time.time()
        for i in range(fill):
            if cfar:  # the oracle's detector (cfar.cpp:27-87; min_distance 2.5 as radar_driver.cpp:52-56 passes it), then the fuser on its cloud
                e = fu.process_cloud(oracle.cfar(buf[i, 0], float(np.float32(kw["range_res"])), float(kw["z_min"]), 2.5, prefix=cfar["window_size"] > 100, **cfar))  # (long windows: the prefix-sum twin, tests/test_cfar_cpu.py)
            else:
                e = fu.process_polar(buf[i, 0])
            S = fu.last_summary()
            no = max(int(S.outer_iterations), 0)
            exp = (int(S.outer_iterations), [int(v) for v in S.inner_iterations[:min(no, 8)]], int(S.num_residuals), int(fu.num_keyframes),
                   len(fu.last_cells()))
            r = rec[i]
            got = (int(r["outer_iterations"]), [int(v) for v in r["inner_iterations"][:min(max(int(r["outer_iterations"]), 0), 8)]],
                   int(r["num_residuals"]), int(r["n_keyframes"]), int(r["n_cells"]))
            t = base + i
            if t > 0 and got != exp:
                mism.append((t, "counts", got, exp))
            g = r["pose"]
            if not (np.all(np.abs(g[:2] - e[:2]) < 1e-4) and abs(g[2] - e[2]) < 1e-5):
                mism.append((t, "pose", [float(v) for v in g], [float(v) for v in e]))
            dev_poses.append(np.array(g)); cpu_poses.append(np.array(e)); cells.append(exp[4]); nkfs.append(exp[3])
        t_cpu += time.time() - t0
        base += fill
        fill = 0
        if log:
            log("%s: %d / %d sweeps, %d disagreements, device %.1f s, oracle %.1f s, wall %.1f s" % (kind, base, T, len(mism), t_dev, t_cpu, time.time() - t_start))

    for t0, chunk in synth.drive_chunks(T, kind, world_seed, seed, A, R, RR, ccw=bool(kw.get("radar_ccw", 0)), procs=procs):
        for img in chunk:
            buf[fill, 0] = img
            fill += 1
            if fill == piece:
                flush()
    flush()
    ctx.pinned_free(buf)
    odo.release()
    ctx.close()
    dev_poses, cpu_poses = np.array(dev_poses), np.array(cpu_poses)
    gtk = kitti.poses_from_xyt(gt)
    out = dict(mismatches=mism, poses_dev=dev_poses, poses_cpu=cpu_poses, gt=gt, cells=np.array(cells), motions=motions,
               seconds_device=t_dev, seconds_oracle=t_cpu, keyframes_max=int(max(nkfs)) if nkfs else 0,
               drift_dev=kitti.drift(gtk, kitti.poses_from_xyt(dev_poses)), drift_cpu=kitti.drift(gtk, kitti.poses_from_xyt(cpu_poses)))
    return out


def run_batched(oracle, params, kind, T, B=3, A=400, R=3360, rr=np.float32(0.0595238), persistent_max=0, route="step", max_cells=None, stats=None, large_kernel=None):
    """B sequences (different drives) through the batched route - cfear_odometry_step_host, or cfear_odometry_replay_host with the
    persistent workgroups switched off (two launches per sweep) - against B oracle fusers, every sweep. stats: optional dict that
    receives the largest cell / residual counts seen (what a test asserts to know which code path it drove)"""
    kw = dict(BASE, range_res=rr)
    kw.update(params)
    fus = [oracle.Fuser(oracle.default_params(**kw)) for _ in range(B)]
    ctx = capi.Context(capi.default_params(**kw), A, R)
    ctx.tune(capi.TUNE_REPLAY_PERSISTENT_MAX, persistent_max)
    odo = ctx.odometry(B, max_cells=max_cells, large_kernel=large_kernel)
    cmax = rmax = 0
    paths = set()
    gens = [synth.drive_chunks(T, kind, 10 + q, 20 + q, A, R, rr, ccw=False) for q in range(B)]
    frames = np.empty((T, B, A, R), dtype=np.uint8)
    for q, g in enumerate(gens):
        for t0, chunk in g:
            frames[t0:t0 + len(chunk), q] = chunk
    recs = None
    if route == "replay":
        recs = odo.replay_host(frames)
    kmax = 0
    for t in range(T):
        if route == "step":
            odo.step_host(frames[t])
            got = odo.poses()
        for q in range(B):
            exp = fus[q].process_polar(frames[t, q])
            So = fus[q].last_summary()
            no = max(int(So.outer_iterations), 0)
            e = (int(So.outer_iterations), [int(v) for v in So.inner_iterations[:min(no, 8)]], int(So.num_residuals), int(fus[q].num_keyframes), len(fus[q].last_cells()))
            if route == "step":
                S, nc, nk = odo.summary(q)
                g = (int(S.outer_iterations), [int(v) for v in S.inner_iterations[:min(max(int(S.outer_iterations), 0), 8)]], int(S.num_residuals), nk, nc)
                pose = got[q]
                if t > 0:
                    paths.add(int(S.assoc_path))
            else:
                r = recs[t, q]
                g = (int(r["outer_iterations"]), [int(v) for v in r["inner_iterations"][:min(max(int(r["outer_iterations"]), 0), 8)]], int(r["num_residuals"]),
                     int(r["n_keyframes"]), int(r["n_cells"]))
                pose = r["pose"]
            if t > 0:
                assert g == e, (t, q, g, e)
            assert np.all(np.abs(pose[:2] - exp[:2]) < 1e-4) and abs(pose[2] - exp[2]) < 1e-5, (t, q, pose, exp)
            kmax = max(kmax, e[3]); cmax = max(cmax, e[4]); rmax = max(rmax, e[2])
    if stats is not None:
        stats.update(cells_max=cmax, residuals_max=rmax, keyframes_max=kmax, assoc_paths=sorted(paths))
    odo.release()
    ctx.close()
    return kmax
