#!/bin/bash
# average duration of the three step kernels in the default bench (rocprofv3 kernel trace); run on the GPU box
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --stream-steps 0 > /tmp/kt.log 2>&1
cd $GRAFT_REPO_ROOT; python tools/rocpd_summary.py $(find /tmp/kt -name "*.db" | head -1) | grep -E "step_kernel|kstrongest" | awk -F'|' '{printf "%s %s us\n", $2, $5}'
tail -1 /tmp/kt.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('scans/s', round(d['value']), 'ms/step', round(d['ms_per_step'],4))"
