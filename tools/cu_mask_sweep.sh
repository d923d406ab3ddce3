#!/bin/bash
# filter on F masked compute units one sweep ahead, odometry on the others: end-to-end rate for a few F (tools/ab_bench.sh output format)
for cfg in ${CFGS:-"0 0" "1 64" "1 80" "1 96" "1 112" "1 128" "2 96" "2 112"}; do
  set -- $cfg
  CFEAR_BENCH_OVERLAP=$1 CFEAR_BENCH_FILTER_CUS=$2 python bench.py --gpus 1 --steps 20 --warmup 5 --repeats 3 --no-presets --no-cpu-baseline --stream-steps 0 --single-sequence-sweeps 0 --no-isolated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('overlap $1 filter_cus $2: value %.0f ms/step %.3f filter %.1f feat %.1f reg %.1f' % (d['value'], d['ms_per_step'], k['kstrongest_launch_us'], k['features_launch_us'], k['registration_launch_us']), ['%.0f' % v for v in d['repeats']['values']])"
done
