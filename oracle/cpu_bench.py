"""CPU baseline of bench.py (TEST INFRASTRUCTURE ONLY): the single-thread oracle run as one process per host core.

  python -m oracle.cpu_bench --procs P --seconds S      -> one JSON line {"value": scans/s over all processes, ...}

Each worker is a fresh interpreter that never loads the HIP runtime: it generates its own synthetic sequence
(same generator and parameters as bench.py, seed = worker index), replays it through oracle/cfear_oracle.c
(cfo_fuser_*) until S seconds have passed, and reports sweeps done / elapsed. Mirrors the reference's own way of
using a host (one offline_odometry process per core, utils/start_workers NR_WORKERS).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(seed, seconds, frames, params_json):
    import numpy as np
    from oracle import binding as ob
    from cfear_radarodometry_code_public_amd import synth
    kw = json.loads(params_json)
    A, R = kw.pop("A"), kw.pop("R")
    kw["range_res"] = np.float32(kw["range_res"])
    p = ob.default_params(**kw)
    imgs, _ = synth.world_sequence(frames, A, R, kw["range_res"], seed=seed, world_seed=1234 + seed, ccw=True, t0=17 * seed)
    f = ob.Fuser(p)
    f.process_polar(imgs[0])  # page everything in before the clock starts
    done, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        f = ob.Fuser(p)
        for t in range(frames):
            f.process_polar(imgs[t])
        done += frames
    print(json.dumps({"done": done, "seconds": time.perf_counter() - t0}), flush=True)


def run(procs, seconds, frames, params):
    """-> dict for bench.py's cpu_baseline; procs worker processes side by side."""
    pj = json.dumps(params)
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--worker"]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    ps = [subprocess.Popen(cmd + [str(i), "--seconds", str(seconds), "--frames", str(frames), "--params", pj], cwd=ROOT, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for i in range(procs)]
    rates, done = [], 0
    for p in ps:
        out, err = p.communicate(timeout=seconds * 10 + 300)
        if p.returncode != 0:
            raise RuntimeError("cpu_bench worker failed: " + err[-400:])
        r = json.loads(out.strip().splitlines()[-1])
        rates.append(r["done"] / r["seconds"]); done += r["done"]
    return {"value": sum(rates), "unit": "scans/s", "cores": procs, "kind": "port",
            "per_core": sum(rates) / len(rates),
            "sample": "%d synthetic 400x3360 sweeps: %d processes (one per host core, %d available) x %d-frame sequences replayed for %.0f s each, "
                      "oracle/cfear_oracle.c single-threaded" % (done, procs, len(os.sched_getaffinity(0)), frames, seconds)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--worker", type=int, default=None)
    ap.add_argument("--procs", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--params", type=str, default="")
    a = ap.parse_args()
    if a.worker is not None:
        worker(a.worker, a.seconds, a.frames, a.params)
        return
    procs = a.procs or len(os.sched_getaffinity(0))
    params = json.loads(a.params) if a.params else dict(A=400, R=3360, range_res=0.0595238, z_min=60.0, min_distance=2.5, k_strongest=12, res=3.0,
                                                        weight_intensity=1, weight_opt=4, cost=1, loss=1, loss_limit=0.1, submap_scan_size=4,
                                                        min_keyframe_dist=1.5, compensate=1, radar_ccw=1)
    print(json.dumps(run(procs, a.seconds, a.frames, params)))


if __name__ == "__main__":
    main()
