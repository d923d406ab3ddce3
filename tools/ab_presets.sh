#!/bin/bash
# A/B on one box of the preset legs (P2D, k = 40 P2P, CFEAR-2, dense world) and the main region: product library vs variants
for round in 1 2; do
  for lib in "" $@; do
    if [ -n "$lib" ]; then export CFEAR_HIP_LIB=$lib; else unset CFEAR_HIP_LIB; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --repeats 2 --no-cpu-baseline --stream-steps 0 --single-sequence-sweeps 0 --no-isolated 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); k=d['kernels']
print(os.environ.get('CFEAR_HIP_LIB','product'), 'main %.0f reg %.1f |' % (d['value'], k['registration_launch_us']), ' | '.join('%s %.0f feat %.0f reg %.0f' % (n, v['scans_per_s'], v['features_launch_us'], v['registration_launch_us']) for n, v in d['presets'].items()))"
  done
done
