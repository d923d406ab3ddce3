"""KITTI drift metric: Python twin vs the C++ tool, and known-answer trajectories."""
import json
import os
import subprocess

import numpy as np

from cfear_radarodometry_code_public_amd import kitti

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "cfear_radarodometry_code_public_amd", "host")


def trajectory(n=2500, step=1.0, yaw_rate=0.004):
    xyt = np.zeros((n, 3))
    for i in range(1, n):
        th = xyt[i - 1, 2]
        xyt[i] = [xyt[i - 1, 0] + step * np.cos(th), xyt[i - 1, 1] + step * np.sin(th), th + yaw_rate]
    return xyt


def test_identical_trajectories_have_zero_drift():
    T = kitti.poses_from_xyt(trajectory())
    d = kitti.drift(T, T)
    assert d["segments"] > 100 and d["translation_percent"] < 1e-9 and d["rotation_deg_per_100m"] < 1e-6


def test_scale_error_is_the_translation_drift():
    xyt = trajectory(yaw_rate=0.0)  # straight: chord length == path length
    gt = kitti.poses_from_xyt(xyt)
    est_xyt = xyt.copy()
    est_xyt[:, :2] *= 1.01  # 1 % scale error, same headings
    d = kitti.drift(gt, kitti.poses_from_xyt(est_xyt))
    assert abs(d["translation_percent"] - 1.0) < 0.02 and d["rotation_deg_per_100m"] < 1e-6


def test_heading_drift_is_the_rotation_drift():
    gt_xyt = trajectory(yaw_rate=0.0)
    est_xyt = trajectory(yaw_rate=1e-4)  # 1e-4 rad per metre = 0.573 deg / 100 m
    d = kitti.drift(kitti.poses_from_xyt(gt_xyt), kitti.poses_from_xyt(est_xyt))
    # a segment ends at the first pose *beyond* its nominal length, so the error is up to one step (1 %) larger
    assert 0.0 <= d["rotation_deg_per_100m"] - np.degrees(1e-4) * 100.0 < 0.01 * np.degrees(1e-4) * 100.0
    assert d["translation_percent"] > 0.1  # the lateral error the heading drift accumulates


def test_text_format_and_cpp_tool_agree(tmp_path):
    rng = np.random.default_rng(3)
    xyt = trajectory(1500)
    gt = kitti.poses_from_xyt(xyt)
    noisy = xyt + np.cumsum(rng.normal(0, [0.01, 0.01, 2e-4], size=xyt.shape), axis=0)
    est = kitti.poses_from_xyt(noisy)
    kitti.write_kitti(tmp_path / "gt.txt", gt)
    kitti.write_kitti(tmp_path / "est.txt", est)
    first = open(tmp_path / "gt.txt").readline().split()
    assert len(first) == 12 and all(len(v.split(".")[1]) == 6 for v in first)  # MatToString: fixed, 6 decimals (types.cpp:64-73)
    back = kitti.read_kitti(tmp_path / "est.txt")
    assert np.abs(back - est).max() < 1e-6
    subprocess.check_call(["make", "-C", HOST, "eval_kitti"], stdout=subprocess.DEVNULL)
    out = subprocess.run([os.path.join(HOST, "eval_kitti"), str(tmp_path / "gt.txt"), str(tmp_path / "est.txt")], capture_output=True, text=True, check=True)
    cpp = json.loads(out.stdout)
    py = kitti.drift(kitti.read_kitti(tmp_path / "gt.txt"), back)
    assert cpp["segments"] == py["segments"] > 0
    assert abs(cpp["translation_percent"] - py["translation_percent"]) < 1e-5  # the tool prints 6 decimals
    assert abs(cpp["rotation_deg_per_100m"] - py["rotation_deg_per_100m"]) < 1e-5
