"""Builds libcfear_hip.so (hand-written HIP, gfx950 only) in-tree with hipcc."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcfear_hip.so")
if os.environ.get("CFEAR_HIP_LIB"):  # tools/: a profile build of the same sources (e.g. tools/build_stop_variants.sh)
    LIB = os.path.abspath(os.environ["CFEAR_HIP_LIB"])

# -ffp-contract=off: several decisions on the path are rounding sensitive (voxel index, float
# d^2 < r^2, float centroid sums) and the reference is built without FMA contraction.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "cfear_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile if a source is newer than the library. Safe to call from several processes at once (one rank per GPU):
    an exclusive file lock serialises them, the later ones find the library up to date."""
    import fcntl
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force, verbose):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for s in sources():
        o = s[:-4] + ".o"
        cmd = ["hipcc"] + FLAGS + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    fail = False
    for s, p in procs:
        out, _ = p.communicate()
        if out and (verbose or p.returncode != 0):
            sys.stderr.write(out.decode())
        if p.returncode != 0:
            fail = True
    if fail:
        raise RuntimeError("hipcc failed")
    tmp = LIB + ".tmp"  # linked beside the target and renamed: nobody ever maps a half-written library
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="-f" in sys.argv, verbose=True))
