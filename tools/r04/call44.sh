#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
for f in 1 2 3 4 5 6 3; do python bench.py --no-e2e --no-cpu-baseline --no-pcie --steps 200 --frames-in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('frames in flight $f:', d['value'], 'Mpx/s', d['ms_per_step'], 'ms')"; done | tee $O/r04_frames_in_flight_settled.txt
for f in 2 3 4; do python bench.py --mix real4k --no-e2e --no-cpu-baseline --no-pcie --steps 200 --frames-in-flight $f 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('real mix, frames in flight $f:', d['value'], 'Mpx/s', d['ms_per_step'], 'ms')"; done | tee -a $O/r04_frames_in_flight_settled.txt
