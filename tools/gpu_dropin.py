"""Drop-in route (host/offline_odometry through the mirror classes) on one synthetic drive: rate + per-call breakdown.
usage: python tools/gpu_dropin.py [sweeps] > profiles/r06_dropin_phases.txt"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cfear_radarodometry_code_public_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
A, R, rr = 400, 3360, np.float32(0.0595238)
imgs, gt = synth.world_sequence(min(n, 64), A, R, rr, seed=11)
T = imgs.shape[0]
idx = [(i % (2 * (T - 1))) for i in range(n)]
idx = [m if m < T else 2 * (T - 1) - m for m in idx]
host = os.path.join(ROOT, "cfear_radarodometry_code_public_amd", "host")
subprocess.check_call(["make", "-C", host], stdout=subprocess.DEVNULL)
with tempfile.TemporaryDirectory() as td:
    f = os.path.join(td, "sweeps.u8")
    imgs[idx].tofile(f)
    for rep in range(3):  # (the third run: every sweep read into one page-locked buffer - a reader that owns its frame buffer - instead of pageable memory)
        r = subprocess.run([os.path.join(host, "offline_odometry"), "--frames", f, "--azimuths", str(A), "--bins", str(R), "--range-res", str(float(rr)), "--res", "3.0",
                            "--submap_scan_size", "4", "--z-min", "60", "--weight_option", "4", "--est_directory", td, "--pinned_frames", "1" if rep == 2 else "0"] + sys.argv[2:], capture_output=True, text=True)
        lines = r.stdout.splitlines()
        print("run %d rc %d%s" % (rep, r.returncode, " (--pinned_frames 1)" if rep == 2 else "")); print("\n".join(l for l in lines if "Frame: %d," % n in l or not l.startswith("Frame")))
        if r.returncode: print(r.stderr)
