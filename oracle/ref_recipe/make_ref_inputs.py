"""Writes oracle/_ref/ref_inputs.bin: the inputs of tests/golden/make_golden.py (same generators, same seeds) in the container
dump_ref_golden.cpp reads. Runs anywhere (numpy only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402
import refio  # noqa: E402
from cfear_radarodometry_code_public_amd import synth  # noqa: E402


def main():
    out = {}
    for name, img in make_golden.small_tiles().items():
        out["tile_" + name] = img
    imgs, gt = synth.world_sequence(8, 400, 3360, make_golden.RR, seed=21)
    for t in range(8):
        out["sweep_%d" % t] = imgs[t]
    out["world_gt"] = gt
    # (k, z_min) per tile run; parameters of the world run (BASELINE configs[1] / [2]); motion of the compensation fixture
    out["tile_kz"] = np.array([[12, 60], [5, 0], [40, 61]], dtype=np.int32)
    out["world_params"] = np.array([float(make_golden.RR), 2.5, 12, 60.0, 3.0, 4, 0.1, 1.5], dtype=np.float64)  # range_res min_distance k z_min res submap loss_limit min_keyframe_dist
    out["comp_motion"] = np.array([1.0, 0.01, 0.02])
    # first guesses of the direct n_scan_normal_reg runs: the ground truth of sweeps 0..3, the last one pushed off by 12 cm / 4 mrad
    rp = gt[:4].copy()
    rp[3] += [0.12, -0.07, 0.004]
    out["reg_poses"] = rp
    out["cfar_params"] = np.array([10, 20, 0.01, 400.0])  # window_size nb_guard_cells false_alarm_rate max_distance
    os.makedirs(os.path.join(ROOT, "oracle", "_ref"), exist_ok=True)
    refio.write(os.path.join(ROOT, "oracle", "_ref", "ref_inputs.bin"), out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
