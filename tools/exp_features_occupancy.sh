cd /tmp && export TMPDIR=/tmp
CFEAR_BENCH_K=6 rocprofv3 --kernel-trace --stats -d /tmp/pf -o b -- python /root/repo/bench.py --no-cpu-baseline > /tmp/b.log 2>&1
tail -1 /tmp/b.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
cd /root/repo; python tools/rocpd_summary.py $(find /tmp/pf -name "*.db" | head -1) | grep -E "step_kernel|kstrongest"
