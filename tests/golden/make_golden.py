"""Generates the golden fixtures in this directory FROM THE CPU ORACLE (oracle/cfear_oracle.c).

The reference has no tests or golden vectors for this path and cannot be built here (ROS, PCL,
Ceres, Eigen, OpenCV absent), so parity is pinned to the oracle's restatement ("parity unpinned"
w.r.t. the reference binary). Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from cfear_radarodometry_code_public_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

RR = np.float32(0.0595238)


def small_tiles():
    rng = np.random.default_rng(2024)
    tiles = {
        "uniform_16x257": rng.integers(0, 256, size=(16, 257), dtype=np.uint8),
        "ties_12x333": synth.ties_scan(12, 333, seed=9),
        "ties2_12x96": synth.ties_scan(12, 96, seed=10, levels=(60, 61)),
        "sparse_8x400": (rng.random((8, 400)) < 0.01).astype(np.uint8) * rng.integers(60, 256, size=(8, 400), dtype=np.uint8),
        "const255_4x128": np.full((4, 128), 255, dtype=np.uint8),
    }
    tiles["edges_6x200"] = rng.integers(0, 100, size=(6, 200), dtype=np.uint8)
    tiles["edges_6x200"][:, :4] = 250
    tiles["edges_6x200"][:, -4:] = 251
    return tiles


def main():
    out = {}
    for name, img in small_tiles().items():
        out["tile_" + name] = img
        for k, z in ((12, 60), (5, 0), (40, 61)):
            out["slots_%s_k%d_z%d" % (name, k, z)] = ob.filter_polar(img, z, k)
    # synthetic-world sequence: the filtered (uncompensated) clouds are the fixture inputs (the polar
    # sweeps themselves are 1.3 MB each and are not committed), then cells / registration trace / trajectory
    imgs, gt = synth.world_sequence(8, 400, 3360, RR, seed=21)
    out["world_gt"] = gt
    kw = dict(range_res=RR, k_strongest=12, z_min=60.0, res=3.0, weight_intensity=1, weight_opt=4, submap_scan_size=4)
    clouds = [ob.cloud(ob.filter_polar(imgs[t], 60, 12), RR, 2.5) for t in range(imgs.shape[0])]
    for t, c in enumerate(clouds):
        out["world_cloud_%d" % t] = c
    for cost, tag in ((1, "p2l"), (2, "p2d")):
        p = ob.default_params(cost=cost, **kw)
        f = ob.Fuser(p)
        traj, iters, ncells = [], [], []
        for t in range(imgs.shape[0]):
            traj.append(f.process_cloud(clouds[t]))
            S = f.last_summary()
            iters.append([S.outer_iterations] + list(S.inner_iterations[:8]))
            ncells.append(len(f.last_cells()))
        out["traj_" + tag] = np.asarray(traj)
        out["iters_" + tag] = np.asarray(iters, dtype=np.int32)
        out["ncells_" + tag] = np.asarray(ncells, dtype=np.int32)
    slots = ob.filter_polar(imgs[3], 60, 12)
    xyi = ob.cloud(slots, RR, 2.5)
    out["world3_slots"] = slots
    out["world3_cloud"] = xyi
    out["world3_cloud_comp"] = ob.compensate(xyi, [1.0, 0.01, 0.02], 0)
    cells = ob.Scan(out["world3_cloud_comp"], ob.default_params(**kw)).cells()
    for f_ in ("mean", "cov", "normal", "lambda_min", "lambda_max", "scale", "nsamples"):
        out["world3_cells_" + f_] = cells[f_]
    np.savez_compressed(os.path.join(HERE, "oracle_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
