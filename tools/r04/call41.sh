#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
echo "-- with the device settle phase (default 60 ms)" | tee -a $O/r04_short_runs.txt
for args in "--steps 20 --warmup 5" "--steps 20 --warmup 5" "--steps 20 --warmup 3" "--steps 100 --warmup 10" "--steps 20 --warmup 5 --settle-ms 0" "--steps 20 --warmup 5 --settle-ms 30" "--steps 20 --warmup 5 --settle-ms 120" "--steps 20 --warmup 5 --config c1" "--steps 20 --warmup 5 --config c5"; do
  python bench.py $args --no-e2e --no-cpu-baseline --no-pcie 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$args:', d['value'], d['ms_per_step'], 'one frame in flight:', (d.get('one_frame_in_flight') or {}).get('value'), d['config']['device_settle_steps'])"
done | tee -a $O/r04_short_runs.txt
