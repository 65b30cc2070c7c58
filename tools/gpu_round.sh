#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, bench, rocprofv3 kernel stats.
# Usage: tools/gpu_round.sh [tag] [bench args...]
tag=${1:-run}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu_$tag.log
tail -4 $O/pytest_gpu_$tag.log
timeout 600 python bench.py "$@" > $O/bench_$tag.log 2>&1
tail -1 $O/bench_$tag.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $R/bench.py --no-cpu-baseline "$@" > $O/prof_$tag.log 2>&1
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/kernel_stats_$tag.csv && head -12 $O/kernel_stats_$tag.csv | cut -c1-160
tail -1 $O/prof_$tag.log | cut -c1-400 > $O/bench_under_rocprof_$tag.log
