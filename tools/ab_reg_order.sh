#!/bin/bash
# A/B on one GPU box: registration workgroups in sequence order (0) against longest first (1, cfear_tune REGISTRATION_ORDER), the driver's
# bench command without the legs after the timed regions, interleaved. $1 = output file.
R=$GRAFT_REPO_ROOT; OUT=${1:-$R/gpurun_out/ab_reg_order.txt}; cd $R; : > $OUT
for o in 0 1 0 1 0 1; do
  CFEAR_BENCH_REG_ORDER=$o timeout 300 python bench.py --no-presets --single-sequence-sweeps 0 --stream-steps 0 --no-cpu-baseline --no-isolated --repeats 3 2>/dev/null > /tmp/ab_reg.json
  python - $o >> $OUT <<'PY'
import json, sys
r = json.load(open("/tmp/ab_reg.json"))
k = r["kernels"]
print("order %s  %.0f scans/s  step %.3f ms  registration %.1f us  features %.1f us  filter %.1f us (per launch of %d)" % (
    sys.argv[1], r["value"], r["ms_per_step"], k["registration_launch_us"], k["features_launch_us"], k["kstrongest_launch_us"], r["config"]["sequences_per_gpu"]))
PY
done
cat $OUT
