#!/bin/bash
# rows walked per filter wave, measured inside the bench's timed region (filter between the odometry kernels, distinct sweeps per sequence)
for round in 1 2; do for r in 4 6 8 12; do
  CFEAR_BENCH_FILTER_ROWS=$r python bench.py --gpus 1 --steps 20 --warmup 5 --repeats 3 --no-presets --no-cpu-baseline --stream-steps 0 --single-sequence-sweeps 0 --no-isolated $@ 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('rows $r: value %.0f filter %.1f (frac %.3f) feat %.1f reg %.1f B %d' % (d['value'], k['kstrongest_launch_us'], d['roofline']['frac'], k['features_launch_us'], k['registration_launch_us'], d['config']['sequences_per_gpu']))"
done; done
