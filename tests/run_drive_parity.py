"""Driving-like replay parity (tests/drive_parity.py) with a log of every disagreement: python tests/run_drive_parity.py kind sweeps [out.json] [preset]
preset: one of PRESETS below (the reference's large-submap settings, tests/test_large_submap_gpu.py), default = BASELINE configs[1]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]  # (a checker script: it lives in tests/ because it runs the oracle)


S10 = dict(k_strongest=40, cost=0, loss=2, loss_limit=0.1, submap_scan_size=10, res=3.0, weight_intensity=1, weight_opt=4, regularization=0.1, covar_scale=1.0)
PRESETS = {"s10_p2p": S10, "s10_p2d": dict(S10, cost=2), "s50_cfear3": dict(S10, submap_scan_size=50),
           "s8_p2l": dict(k_strongest=12, cost=1, loss=1, loss_limit=0.1, res=3.0, weight_intensity=0, weight_opt=0, submap_scan_size=8),
           # round 5: the reference's evaluation grid (tests/test_eval_grid_gpu.py) on long drives
           "cfear1": dict(cost=1, submap_scan_size=1, res=3.5, k_strongest=12, loss=1, loss_limit=0.1, covar_scale=1.0, regularization=1.0, weight_intensity=0, weight_opt=4),
           "nocomp_p2p_k40": dict(cost=0, submap_scan_size=4, res=3.0, k_strongest=40, loss=1, loss_limit=0.1, weight_intensity=1, weight_opt=0, compensate=0),
           "res1_s3": dict(cost=1, submap_scan_size=3, res=1.0, k_strongest=12, loss=1, loss_limit=0.1, weight_intensity=1, weight_opt=0),
           # params/kstrong_vs_cfar/oxford-cfear-3-ca-cfar: the detector in front of the fuser (tests/test_cfar_odometry_gpu.py)
           # round 6: CFEAR-3 as shipped (launch/oxford_demo:32-40: P2P, k = 40, four keyframes), for the street world (profiles/r06_world_realism.json)
           "cfear3_k40_p2p": dict(k_strongest=40, cost=0, submap_scan_size=4, res=3.0, loss=1, loss_limit=0.1, weight_intensity=1, weight_opt=4),
           "ca_cfar_w500": dict(z_min=20.0, cost=0, submap_scan_size=4, res=3.0, loss=1, loss_limit=0.1, weight_intensity=0, weight_opt=0, regularization=1.0, covar_scale=1.0,
                                cfar=dict(window_size=500, nb_guard_cells=10, false_alarm_rate=0.0001)),
           "ca_cfar": dict(z_min=20.0, cost=0, submap_scan_size=4, res=3.0, loss=1, loss_limit=0.1, weight_intensity=0, weight_opt=0, regularization=1.0, covar_scale=1.0,
                           cfar=dict(window_size=40, nb_guard_cells=10, false_alarm_rate=0.01))}


def main():
    import numpy as np
    import drive_parity
    from oracle import binding
    kind, T = sys.argv[1], int(sys.argv[2])
    pm = os.environ.get("CFEAR_REPLAY_PERSISTENT_MAX")
    preset = sys.argv[4] if len(sys.argv) > 4 else None
    out = drive_parity.run(binding, T, kind, log=lambda s: print(s, flush=True), persistent_max=int(pm) if pm else None,
                           piece=int(os.environ.get("CFEAR_REPLAY_PIECE", "250")), params=PRESETS[preset] if preset else None)
    reg = drive_parity.regimes(out["motions"])
    bad = sorted(set(m[0] for m in out["mismatches"]))
    rep = {"kind": kind, "sweeps": T, "preset": preset or "configs[1]", "keyframes_max": out["keyframes_max"], "disagreeing_sweeps": len(bad), "first": [list(map(str, m)) for m in out["mismatches"][:20]],
           "by_regime": {k: int(np.sum(v[bad])) if bad else 0 for k, v in reg.items()}, "regime_sizes": {k: int(v.sum()) for k, v in reg.items()},
           "cells_median": float(np.median(out["cells"])), "seconds_device": out["seconds_device"], "seconds_oracle": out["seconds_oracle"],
           "sweeps_per_s_device": T / out["seconds_device"], "drift_dev": out["drift_dev"], "drift_cpu": out["drift_cpu"]}
    print(json.dumps(rep, indent=1))
    if len(sys.argv) > 3:
        json.dump(rep, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
