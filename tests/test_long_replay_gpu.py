"""Long-horizon parity = BASELINE configs[4] proxy (full recording replay + KITTI drift against the CPU run,
offline_odometry.cpp:73-141, eval_trajectory.cpp:169-232): a 1000-sweep synthetic recording at the Oxford shape (400 x 3768,
range_res 0.0438) is written as a rosbag, replayed through replay.py on the device and through the oracle's fuser on the CPU.
Hundreds of keyframe turnovers: the keyframe count, the outer and inner iteration counts of every Register call, every
pose (1e-4 m / 1e-5 rad) and the KITTI drift (1e-6) must agree."""
import os

import numpy as np
import pytest

from cfear_radarodometry_code_public_amd import kitti, readers, replay, synth

pytestmark = pytest.mark.gpu
T = int(os.environ.get("CFEAR_LONG_SWEEPS", "1000"))
A, R, RR = 400, 3768, np.float32(0.0438)


def test_1000_sweep_bag_replay_matches_oracle_at_every_sweep(oracle, tmp_path):
    chunks, gt = synth.world_sequence_long(T, A, R, RR, seed=71, world_seed=4321, ccw=False, chunk=25)
    bag = tmp_path / "long.bag"
    w = readers.BagWriter(bag, compression="none")
    i = 0
    for ch in chunks:
        for img in ch:
            t = 1547131046000000000 + i * 250000000
            w.write("/gt", "nav_msgs/Odometry", t, readers.encode_odometry(gt[i] + [3.0, 4.0, 0.0], t, seq=i))
            w.write("/Navtech/Polar", "sensor_msgs/Image", t + 500, readers.encode_image(np.asarray(img), t + 500, seq=i))
            i += 1
    w.close()
    assert i == T
    out = replay.main(["--bag", str(bag), "--est_directory", str(tmp_path / "est"), "--range-res", "0.0438", "--res", "3.0", "--z-min", "60",
                       "--submap_scan_size", "4", "--weight_option", "4", "--trace"])
    assert out["frames"] == T and len(out["trace"]) == T
    # the CPU run of the same recording (checker): oracle fuser, sweep by sweep
    fu = oracle.Fuser(oracle.default_params(range_res=RR, z_min=60.0, res=3.0, submap_scan_size=4, weight_opt=4, weight_intensity=1, compensate=1,
                                            radar_ccw=0, cost=1, loss=1))
    exp_poses, first_bad = [], None
    i = 0
    for ch in chunks:
        for img in ch:
            e = fu.process_polar(np.asarray(img))
            exp_poses.append(e)
            tr = out["trace"][i]
            if i > 0:
                S = fu.last_summary()
                exp = (int(S.outer_iterations), [int(v) for v in S.inner_iterations[:max(S.outer_iterations, 0)]], int(S.num_residuals),
                       int(fu.num_keyframes))
                got = (tr["outer"], tr["inner"], tr["residuals"], tr["keyframes"])
                if first_bad is None and got != exp:
                    first_bad = (i, got, exp)
            g = out["poses"][i]
            if first_bad is None and not (np.all(np.abs(g[:2] - e[:2]) < 1e-4) and abs(g[2] - e[2]) < 1e-5):
                first_bad = (i, g.tolist(), e.tolist())
            i += 1
    assert first_bad is None, "first disagreement (sweep, device, oracle): %r" % (first_bad,)
    exp_poses = np.array(exp_poses)
    # the sensor moved ~1 m per sweep: a new keyframe every second sweep, so the ring of 4 turned over hundreds of times
    travelled = float(np.sum(np.linalg.norm(np.diff(exp_poses[:, :2], axis=0), axis=1)))
    assert travelled > 0.9 * (T - 1)
    # KITTI files and drift: device trajectory vs the CPU trajectory, both against the recording's ground truth
    est = kitti.read_kitti(tmp_path / "est" / "est_00.txt")
    gtk = kitti.read_kitti(tmp_path / "est" / "gt_00.txt")
    assert est.shape == (T, 4, 4) and gtk.shape == (T, 4, 4)
    d_dev = kitti.drift(gtk, est)
    d_cpu = kitti.drift(gtk, kitti.poses_from_xyt(exp_poses))
    assert d_dev["segments"] == d_cpu["segments"] and d_dev["segments"] > 0
    assert abs(d_dev["translation_percent"] - d_cpu["translation_percent"]) < 1e-3  # est_00.txt carries 6 decimals
    assert abs(d_dev["rotation_deg_per_100m"] - d_cpu["rotation_deg_per_100m"]) < 1e-3
    d_dev_full = kitti.drift(gtk, kitti.poses_from_xyt(out["poses"]))  # full precision poses: 1e-6
    assert abs(d_dev_full["translation_percent"] - d_cpu["translation_percent"]) < 1e-6
    assert abs(d_dev_full["rotation_deg_per_100m"] - d_cpu["rotation_deg_per_100m"]) < 1e-6
    assert abs(out["drift"]["translation_percent"] - d_dev_full["translation_percent"]) < 1e-5  # replay.py: full-precision poses and ground truth (gtk above was read back from 6-decimal text)
    assert d_cpu["translation_percent"] < 5.0  # known answer: the odometry follows the synthetic ground truth


@pytest.mark.parametrize("name,kw,sweeps", [
    ("config2_p2d", dict(cost=2, regularization=0.1, covar_scale=1.0, radar_ccw=1, min_keyframe_dist=1.5), 300),
    ("cfear3_p2p_k40", dict(cost=0, k_strongest=40, submap_scan_size=4), 150),
    ("cauchy_p2l_3_keyframes", dict(cost=1, loss=2, loss_limit=0.2, submap_scan_size=3, res=3.5), 200),
])
def test_long_runs_of_the_other_costs_match_the_oracle_at_every_sweep(oracle, name, kw, sweeps):
    """BASELINE configs[2] (P2D) and the reference's other presets over hundreds of sweeps of the configs[1] stream (400 x 3360):
    the evaluation kernels of these costs / losses are instantiations of their own, and the k = 40 preset runs the general
    cloud and feature paths. Iteration counts, residual counts, keyframe counts and poses at every sweep."""
    from cfear_radarodometry_code_public_amd import capi
    T2 = int(os.environ.get("CFEAR_LONG_SWEEPS_OTHER", str(sweeps)))
    ccw = bool(kw.get("radar_ccw", 0))
    chunks, gt = synth.world_sequence_long(T2, 400, 3360, np.float32(0.0595238), seed=73, world_seed=977, ccw=ccw, chunk=25)
    base = dict(range_res=np.float32(0.0595238), k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, compensate=1,
                radar_ccw=0, cost=1, loss=1, loss_limit=0.1, submap_scan_size=4)
    base.update(kw)
    fu = oracle.Fuser(oracle.default_params(**base))
    ctx = capi.Context(capi.default_params(**base), 400, 3360)
    odo = ctx.odometry(1)
    i, first_bad = 0, None
    for ch in chunks:
        for img in ch:
            img = np.asarray(img)
            odo.step_host(img[None])
            g = odo.poses()[0]
            e = fu.process_polar(img)
            S, nc, nk = odo.summary(0)
            So = fu.last_summary()
            got = (int(S.outer_iterations), [int(v) for v in S.inner_iterations[:8]], int(S.num_residuals), nk, nc)
            exp = (int(So.outer_iterations), [int(v) for v in So.inner_iterations[:8]], int(So.num_residuals), int(fu.num_keyframes), len(fu.last_cells()))
            if first_bad is None and (got != exp or not (np.all(np.abs(g[:2] - e[:2]) < 1e-4) and abs(g[2] - e[2]) < 1e-5)):
                first_bad = (i, got, exp, g.tolist(), e.tolist())
            i += 1
    assert first_bad is None, "first disagreement (sweep, device, oracle, poses): %r" % (first_bad,)
    assert i == T2 and np.linalg.norm(g[:2] - gt[-1, :2]) < 0.02 * T2 + 1.0  # known answer: the odometry follows the synthetic ground truth
    odo.release()
    ctx.close()
