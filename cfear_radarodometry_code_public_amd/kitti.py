"""KITTI odometry drift metric (translation %, rotation deg / 100 m) and KITTI-format trajectory I/O.

Python twin of include/cfear_hip/kitti_metric.hpp (same segment lengths 100..800 m, a segment start every 10 frames, error
= delta_est^-1 * delta_gt per segment). The text format is the one the reference's EvalTrajectory::Write emits
(eval_trajectory.cpp:169-184 via MatToString, types.cpp:64-73): one 3x4 row-major pose per line, fixed, 6 decimals.
"""
import numpy as np

LENGTHS = (100.0, 200.0, 300.0, 400.0, 500.0, 600.0, 700.0, 800.0)
STEP = 10


def poses_from_xyt(xyt):
    """(n, 3) array of (x, y, theta) -> (n, 4, 4) homogeneous poses (planar motion, z = 0)."""
    xyt = np.asarray(xyt, dtype=np.float64)
    T = np.tile(np.eye(4), (len(xyt), 1, 1))
    c, s = np.cos(xyt[:, 2]), np.sin(xyt[:, 2])
    T[:, 0, 0], T[:, 0, 1], T[:, 1, 0], T[:, 1, 1] = c, -s, s, c
    T[:, 0, 3], T[:, 1, 3] = xyt[:, 0], xyt[:, 1]
    return T


def write_kitti(path, poses):
    with open(path, "w") as fh:
        for T in poses:
            fh.write(" ".join("%.6f" % v for v in np.asarray(T)[:3, :4].reshape(-1)) + "\n")


def read_kitti(path):
    rows = np.loadtxt(path, dtype=np.float64, ndmin=2)
    T = np.tile(np.eye(4), (len(rows), 1, 1))
    T[:, :3, :4] = rows.reshape(-1, 3, 4)
    return T


def drift(gt, est):
    """-> dict(translation_percent, rotation_deg_per_100m, segments)"""
    n = min(len(gt), len(est))
    gt, est = np.asarray(gt)[:n], np.asarray(est)[:n]
    if n < 2:
        return {"translation_percent": 0.0, "rotation_deg_per_100m": 0.0, "segments": 0}
    dist = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(gt[:, :3, 3], axis=0), axis=1))])
    t_errs, r_errs = [], []
    for first in range(0, n, STEP):
        for ln in LENGTHS:
            idx = np.nonzero(dist[first:] > dist[first] + ln)[0]
            if len(idx) == 0:
                continue
            last = first + int(idx[0])
            dgt = np.linalg.inv(gt[first]) @ gt[last]
            des = np.linalg.inv(est[first]) @ est[last]
            e = np.linalg.inv(des) @ dgt
            c = min(max(0.5 * (np.trace(e[:3, :3]) - 1.0), -1.0), 1.0)
            r_errs.append(np.arccos(c) / ln)
            t_errs.append(np.linalg.norm(e[:3, 3]) / ln)
    if not t_errs:
        return {"translation_percent": 0.0, "rotation_deg_per_100m": 0.0, "segments": 0}
    return {"translation_percent": 100.0 * float(np.mean(t_errs)),
            "rotation_deg_per_100m": float(np.mean(r_errs)) * 180.0 / np.pi * 100.0, "segments": len(t_errs)}
