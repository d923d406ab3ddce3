#!/bin/bash
# kernel durations of the drop-in route (host/offline_odometry, 300 sweeps) under rocprofv3 --kernel-trace; run on the GPU box
R=$GRAFT_REPO_ROOT
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
from cfear_radarodometry_code_public_amd import synth
imgs, gt = synth.world_sequence(64, 400, 3360, np.float32(0.0595238), seed=11)
idx = [(i % 126) for i in range(300)]; idx = [m if m < 64 else 126 - m for m in idx]
imgs[idx].tofile("/tmp/sweeps.u8")
PY
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- $R/cfear_radarodometry_code_public_amd/host/offline_odometry --frames /tmp/sweeps.u8 --azimuths 400 --bins 3360 --range-res 0.0595238 --res 3.0 --submap_scan_size 4 --z-min 60 --weight_option 4 --est_directory /tmp "$@" > /tmp/kt.log 2>&1
tail -3 /tmp/kt.log | head -1
cd $R; python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1)
