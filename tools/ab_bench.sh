#!/bin/bash
# A/B on one box: the driver's bench command (main region only) against the product library and against variants, interleaved twice
for round in 1 2; do
  for lib in "" $@; do
    if [ -n "$lib" ]; then export CFEAR_HIP_LIB=$lib; else unset CFEAR_HIP_LIB; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-presets --no-cpu-baseline --stream-steps 0 --single-sequence-sweeps 0 --no-isolated 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']
print(os.environ.get('CFEAR_HIP_LIB','product'), 'value %.0f filter %.1f feat %.1f reg %.1f' % (d['value'], k['kstrongest_launch_us'], k['features_launch_us'], k['registration_launch_us']), ['%.0f' % v for v in d['repeats']['values']])"
  done
done
