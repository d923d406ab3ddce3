#!/bin/bash
# VALU / SALU / LDS instructions per azimuth row and kernel time of the k-strongest kernel on the three input families of SURVEY 8(d):
# S-uniform (bandwidth stress), S-world, S-ties. Counters in their own pass (--pmc with --kernel-trace only). $1 = output file.
R=$GRAFT_REPO_ROOT; OUT=${1:-$R/gpurun_out/k1_inputs.txt}; cd /tmp && export TMPDIR=/tmp
N=${K1_N:-1536}
K1_TIES=1 K1_UNIFORM_SEEDED=1 K1_N=$N K1_REPS=1 K1_CONFIGS="7,0" timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d /tmp/pmc_k1_inputs -o k1 -- python $R/tools/gpu_time_k1.py > /tmp/pmc_k1_inputs.log 2>&1
K1_TIES=1 K1_UNIFORM_SEEDED=1 K1_N=$N K1_REPS=3 K1_CONFIGS="7,0" timeout 300 python $R/tools/gpu_time_k1.py > /tmp/k1_inputs_time.log 2>&1
python - "$N" > $OUT <<'PY'
import sqlite3, glob, collections, sys
N = int(sys.argv[1])
db = glob.glob('/tmp/pmc_k1_inputs/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like '%kstrongest%' order by dispatch_id"))
d = collections.defaultdict(list)
for n, i, v in rows: d[n].append((i, v))
names = ["uniform", "world", "ties"]  # the order gpu_time_k1.py launches them in (dict order), 12 launches each (2 warm-up + 10 timed)
print("k-strongest kernel by input family, %d-scan launches (%d azimuth rows), per-row averages over 12 launches each; PMC pass separate from the timing pass" % (N, N * 400))
print("%-18s %10s %10s %10s" % ("counter / row", *names))
for n, l in sorted(d.items()):
    l.sort(); per = 12
    groups = [l[i:i + per] for i in range(0, len(l), per)]
    print("%-18s %s" % (n, " ".join("%10.1f" % (sum(v for _, v in g) / len(g) / (N * 400)) for g in groups[:3])))
print()
print(open('/tmp/k1_inputs_time.log').read())
PY
cat $OUT
