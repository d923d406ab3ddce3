#!/bin/bash
# VALU / SALU / LDS instructions per row of the k-strongest kernel (world data), (uniform, world)
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
K1_N=256 K1_REPS=1 K1_CONFIGS="${1:-7,4}" timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/pmc_valu -o k1 -- python $R/tools/gpu_time_k1.py > /tmp/pmc_valu.log 2>&1
python - <<'PY'
import sqlite3, glob, collections
db = glob.glob('/tmp/pmc_valu/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like '%kstrongest%' order by dispatch_id"))
d = collections.defaultdict(list)
for n, i, v in rows: d[n].append((i, v))
for n, l in sorted(d.items()):
    l.sort(); per = 12  # 2 warm + 10 timed launches per (config, data set)
    groups = [l[i:i + per] for i in range(0, len(l), per)]
    print(n, " ".join("%.1f" % (sum(v for _, v in g) / len(g) / (256 * 400)) for g in groups), "(per row; groups = configs x [uniform, world])")
PY
