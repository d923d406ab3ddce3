#!/bin/bash
# per-phase instruction counts of the k-strongest kernel per azimuth row and input family: PMC counters of the stop-after-phase-n
# builds (tools/build_k1_stop_variants.sh), cumulative; the last line is the product kernel. $1 = output file.
R=$GRAFT_REPO_ROOT; OUT=${1:-$R/gpurun_out/k1_phases.txt}; cd /tmp && export TMPDIR=/tmp
N=${K1_N:-1536}
: > $OUT
echo "k-strongest kernel, cumulative per-row counters of the builds that stop after phase n (1 load + LDS staging, 2 threshold search + candidate masks," >> $OUT
echo "3 candidates -> lanes, 4 ranking, full = + suppression + emit); $N-scan launches, 12 launches per input; columns: uniform world ties" >> $OUT
for n in 1 2 3 4 full; do
  if [ $n = full ]; then unset CFEAR_HIP_LIB; else export CFEAR_HIP_LIB=$R/tools/_stop/libcfear_hip_k1stop$n.so; fi
  rm -rf /tmp/pmc_k1p
  K1_TIES=1 K1_UNIFORM_SEEDED=1 K1_N=$N K1_REPS=1 K1_CONFIGS="7,0" timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d /tmp/pmc_k1p -o k1 -- python $R/tools/gpu_time_k1.py > /tmp/pmc_k1p_$n.log 2>&1
  python - "$N" "$n" >> $OUT <<'PY'
import sqlite3, glob, collections, sys
N, tag = int(sys.argv[1]), sys.argv[2]
db = glob.glob('/tmp/pmc_k1p/**/*.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = list(c.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like '%kstrongest%' order by dispatch_id"))
d = collections.defaultdict(list)
for n, i, v in rows: d[n].append((i, v))
for n, l in sorted(d.items()):
    l.sort(); per = 12
    groups = [l[i:i + per] for i in range(0, len(l), per)]
    print("stop %-5s %-16s %s" % (tag, n, " ".join("%9.1f" % (sum(v for _, v in g) / len(g) / (N * 400)) for g in groups[:3])))
PY
done
cat $OUT
