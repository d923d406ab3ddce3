"""ROS-free replay of a recorded sequence through the device odometry (the reference's offline_odometry.cpp:60-127 loop):

    python -m cfear_radarodometry_code_public_amd.replay --bag radar.bag --est_directory out [--gt_directory out]
    python -m cfear_radarodometry_code_public_amd.replay --oxford_png_dir <sequence>/radar --est_directory out

Reads /Navtech/Polar sweeps (and /gt odometry when present) from a rosbag v2.0 file, or the PNG sweeps of an Oxford Radar
RobotCar sequence in file-name (timestamp) order, hands them to cfear_odometry_replay_host in pieces (one sequence; no host
round trip per sweep) and writes the trajectory in the KITTI text format of EvalTrajectory::Write (est_00.txt, gt_00.txt).
Prints the replay rate (sweeps / second, what offline_odometry.cpp:125 prints) and, with ground truth, the KITTI drift.
Needs a GPU: there is no CPU path.
"""
import argparse
import glob
import json
import os

import numpy as np

from . import capi, kitti, readers


def sweeps(args):
    if args.bag:
        for kind, t, payload in readers.BagReader(args.bag).sweeps_and_gt(args.image_topic, args.gt_topic):
            yield (kind, t, payload if kind == "gt" else readers.polar_image(payload, args.dataset))
    else:
        for path in sorted(glob.glob(os.path.join(args.oxford_png_dir, "*.png"))):
            d = readers.read_oxford_png(path)
            yield ("image", int(d["timestamps"][0]) * 1000, d["polar"])


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    src = ap.add_mutually_exclusive_group(required=True)
    src.add_argument("--bag")
    src.add_argument("--oxford_png_dir")
    ap.add_argument("--dataset", default="oxford")
    ap.add_argument("--image_topic", default="/Navtech/Polar")
    ap.add_argument("--gt_topic", default="/gt")
    ap.add_argument("--est_directory", default=".")
    ap.add_argument("--gt_directory", default=None)
    ap.add_argument("--max_frames", type=int, default=0)
    # defaults of offline_odometry.cpp:155-187
    ap.add_argument("--range-res", dest="range_res", type=float, default=0.0438)
    ap.add_argument("--z-min", dest="z_min", type=float, default=65.0)
    ap.add_argument("--k_strongest", type=int, default=12)
    ap.add_argument("--min_distance", type=float, default=2.5)
    ap.add_argument("--res", type=float, default=3.5)
    ap.add_argument("--submap_scan_size", type=int, default=3)
    ap.add_argument("--weight_intensity", type=int, default=1)
    ap.add_argument("--weight_option", type=int, default=0)
    ap.add_argument("--cost_type", default="P2L", choices=["P2P", "P2L", "P2D"])
    ap.add_argument("--loss_type", default="Huber", choices=["None", "Huber", "Cauchy", "SoftLOne", "Combined", "Tukey"])
    ap.add_argument("--loss_limit", type=float, default=0.1)
    ap.add_argument("--radar_ccw", type=int, default=0)
    ap.add_argument("--disable_compensate", type=int, default=0)
    ap.add_argument("--registered_min_keyframe_dist", type=float, default=1.5)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--piece", type=int, default=256, help="sweeps handed to one cfear_odometry_replay_host call (pinned staging buffer)")
    ap.add_argument("--trace", action="store_true",
                    help="keep per-sweep poses, Register summaries (outer / inner iteration counts, residuals), keyframe and cell counts in the result")
    args = ap.parse_args(argv)
    cost = {"P2P": 0, "P2L": 1, "P2D": 2}[args.cost_type]
    loss = {"None": 0, "Huber": 1, "Cauchy": 2, "SoftLOne": 3, "Combined": 4, "Tukey": 5}[args.loss_type]
    import time
    ctx = odo = buf = None
    est, gts, n, trace = [], [], 0, []
    first_gt = None
    fill = 0
    t_wall0 = time.perf_counter()
    t_dev = 0.0

    def flush():
        # offline_odometry.cpp:103-125 for the sweeps collected so far: one cfear_odometry_replay_host call (no host round trip per
        # sweep; the sweeps sit in pinned memory, so their copies overlap with the kernels of the chunk before)
        nonlocal fill, t_dev
        if fill == 0:
            return
        t0 = time.perf_counter()
        rec = odo.replay_host(buf[:fill])[:, 0]
        t_dev += time.perf_counter() - t0
        for r in rec:
            est.append(np.array(r["pose"]))
            if args.trace:
                no = min(max(int(r["outer_iterations"]), 0), 8)
                trace.append({"outer": int(r["outer_iterations"]), "inner": [int(v) for v in r["inner_iterations"][:no]], "residuals": int(r["num_residuals"]),
                              "final_cost": float(r["final_cost"]), "keyframes": int(r["n_keyframes"]), "cells": int(r["n_cells"])})
        fill = 0

    for kind, t, payload in sweeps(args):
        if kind == "gt":
            x, y, th = payload  # relative to the first ground-truth pose (offline_odometry.cpp:91-92)
            T = kitti.poses_from_xyt([[x, y, th]])[0]
            if first_gt is None:
                first_gt = np.linalg.inv(T)
            gts.append(first_gt @ T)
            continue
        img = payload
        if ctx is None:
            p = capi.default_params(range_res=np.float32(args.range_res), z_min=args.z_min, k_strongest=args.k_strongest, min_distance=args.min_distance,
                                    res=args.res, submap_scan_size=args.submap_scan_size, weight_intensity=args.weight_intensity,
                                    weight_opt=args.weight_option, cost=cost, loss=loss, loss_limit=args.loss_limit, radar_ccw=args.radar_ccw,
                                    compensate=0 if args.disable_compensate else 1, min_keyframe_dist=args.registered_min_keyframe_dist)
            ctx = capi.Context(p, img.shape[0], img.shape[1], device=args.device)
            odo = ctx.odometry(1)
            buf = ctx.pinned((max(1, args.piece), 1, img.shape[0], img.shape[1]))
        buf[fill, 0] = img
        fill += 1
        n += 1
        if fill == buf.shape[0]:
            flush()
        if args.max_frames and n >= args.max_frames:
            break
    if ctx is not None:
        flush()
    t_wall = time.perf_counter() - t_wall0
    if not est:
        raise SystemExit("no radar sweeps found")
    os.makedirs(args.est_directory, exist_ok=True)
    est_T = kitti.poses_from_xyt(np.array(est))
    kitti.write_kitti(os.path.join(args.est_directory, "est_00.txt"), est_T)
    # the rate offline_odometry.cpp:125 prints (frames / second): of the device part alone and of the whole loop incl. reading / decoding
    out = {"frames": n, "final_pose": [float(v) for v in est[-1]], "sweeps_per_s_device": n / t_dev if t_dev > 0 else None,
           "sweeps_per_s_with_reading": n / t_wall if t_wall > 0 else None}
    if gts:
        gdir = args.gt_directory or args.est_directory
        os.makedirs(gdir, exist_ok=True)
        m = min(len(gts), len(est_T))
        kitti.write_kitti(os.path.join(gdir, "gt_00.txt"), gts[:m])
        out["drift"] = kitti.drift(np.array(gts[:m]), est_T[:m])
    print(json.dumps(out))
    if args.trace:
        out["poses"] = np.array(est)
        out["trace"] = trace
    odo.release()
    ctx.close()
    return out


if __name__ == "__main__":
    main()
