"""Driving-like replay parity (tests/drive_parity.py) with a log of every disagreement: python tests/run_drive_parity.py kind sweeps [out.json]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]  # (a checker script: it lives in tests/ because it runs the oracle)


def main():
    import numpy as np
    import drive_parity
    from oracle import binding
    kind, T = sys.argv[1], int(sys.argv[2])
    pm = os.environ.get("CFEAR_REPLAY_PERSISTENT_MAX")
    out = drive_parity.run(binding, T, kind, log=lambda s: print(s, flush=True), persistent_max=int(pm) if pm else None,
                           piece=int(os.environ.get("CFEAR_REPLAY_PIECE", "250")))
    reg = drive_parity.regimes(out["motions"])
    bad = sorted(set(m[0] for m in out["mismatches"]))
    rep = {"kind": kind, "sweeps": T, "disagreeing_sweeps": len(bad), "first": [list(map(str, m)) for m in out["mismatches"][:20]],
           "by_regime": {k: int(np.sum(v[bad])) if bad else 0 for k, v in reg.items()}, "regime_sizes": {k: int(v.sum()) for k, v in reg.items()},
           "cells_median": float(np.median(out["cells"])), "seconds_device": out["seconds_device"], "seconds_oracle": out["seconds_oracle"],
           "sweeps_per_s_device": T / out["seconds_device"], "drift_dev": out["drift_dev"], "drift_cpu": out["drift_cpu"]}
    print(json.dumps(rep, indent=1))
    if len(sys.argv) > 3:
        json.dump(rep, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
