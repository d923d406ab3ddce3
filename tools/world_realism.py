"""Why does the reference's CFEAR-3 as shipped (P2P, k = 40; launch/oxford_demo:32-40) drift several per cent on the synthetic drives where P2L drifts a
fraction of one?  Checker script: runs the ORACLE's fuser on the CPU (no GPU needed) over a synthetic drive and reports, per configuration, the KITTI
drift against ground truth, how the registered step compares with the true step (along track / across track / yaw), residual counts, inner iterations and
how often the inner solve ends at its 20-iteration limit.

  python tools/world_realism.py [kind=canyon] [sweeps=400] [out.json] [world kwargs as k=v ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cfear_radarodometry_code_public_amd import kitti, synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

A, R, RR = 400, 3768, np.float32(0.0438)
BASE = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, compensate=1, radar_ccw=0, cost=1, loss=1, loss_limit=0.1,
            submap_scan_size=4)
CONFIGS = {"p2l_k12": dict(), "p2p_k40": dict(k_strongest=40, cost=0), "p2l_k40": dict(k_strongest=40), "p2p_k12": dict(cost=0), "p2d_k12": dict(cost=2, regularization=0.1)}


def rel_motion(p):
    d = np.zeros((len(p) - 1, 3))
    c, s = np.cos(p[:-1, 2]), np.sin(p[:-1, 2])
    dx, dy = p[1:, 0] - p[:-1, 0], p[1:, 1] - p[:-1, 1]
    d[:, 0] = c * dx + s * dy; d[:, 1] = -s * dx + c * dy
    d[:, 2] = np.arctan2(np.sin(p[1:, 2] - p[:-1, 2]), np.cos(p[1:, 2] - p[:-1, 2]))
    return d


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "canyon"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    out = sys.argv[3] if len(sys.argv) > 3 else None
    cfgs = os.environ.get("WR_CONFIGS", "p2l_k12,p2p_k40").split(",")
    world = synth.DriveWorld(kind, 0)
    _, motions, gt = synth.drive_plan(T, world, 1)
    imgs = np.empty((T, A, R), dtype=np.uint8)
    t0 = time.time()
    for s0, chunk in synth.drive_chunks(T, kind, 0, 1, A, R, RR, ccw=False):
        imgs[s0:s0 + len(chunk)] = chunk
    res = {"kind": kind, "sweeps": T, "render_s": time.time() - t0, "configs": {}}
    gm = rel_motion(gt)
    moving = np.abs(gm[:, 0]) > 0.5
    for name in cfgs:
        kw = dict(BASE); kw.update(CONFIGS[name])
        fu = ob.Fuser(ob.default_params(**kw))
        poses = np.zeros((T, 3)); nres = []; inner = []; cells = []; pts = []
        for t in range(T):
            poses[t] = fu.process_polar(imgs[t])
            S = fu.last_summary()
            no = max(int(S.outer_iterations), 0)
            nres.append(int(S.num_residuals)); cells.append(len(fu.last_cells()))
            inner.append([int(v) for v in S.inner_iterations[:min(no, 8)]])
        m = rel_motion(poses)
        ratio = m[moving, 0] / gm[moving, 0]
        flat = [v for it in inner[1:] for v in it]
        r = {"drift": kitti.drift(kitti.poses_from_xyt(gt), kitti.poses_from_xyt(poses)),
             "step_ratio_mean": float(np.mean(ratio)), "step_ratio_std": float(np.std(ratio)),
             "across_track_err_rms_m": float(np.sqrt(np.mean((m[:, 1] - gm[:, 1]) ** 2))), "yaw_err_rms_rad": float(np.sqrt(np.mean((m[:, 2] - gm[:, 2]) ** 2))),
             "along_track_err_rms_m": float(np.sqrt(np.mean((m[:, 0] - gm[:, 0]) ** 2))),
             "residuals_median": float(np.median(nres[1:])), "cells_median": float(np.median(cells)),
             "inner_iterations_mean": float(np.mean(flat)), "inner_at_limit_frac": float(np.mean(np.array(flat) >= 21)),
             "end_error_m": float(np.hypot(*(poses[-1, :2] - gt[-1, :2])))}
        res["configs"][name] = r
        print(name, json.dumps(r), flush=True)
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
