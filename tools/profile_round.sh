#!/bin/bash
# GPU box: the round's evidence for one bench configuration -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc_traffic.json}
# usage: tools/profile_round.sh <tag> [bench args]
# (--frames-in-flight 1: one context, so that rocprofv3's per-kernel durations are those of kernels running alone --
# with several frames in flight launches of different streams overlap and the profiler's durations include the wait)
tag=$1; shift
export JXLHIP_BENCH_NO_GRAPH=1  # (profiling / experiment runs: no hipGraph side measurement)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $R/bench.py --no-cpu-baseline --no-pcie --frames-in-flight 1 --steps 50 --warmup 5 "$@" > $O/${tag}_bench.log 2>&1
grep '^{' $O/${tag}_bench.log > $O/${tag}_bench.json
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::native\|rocclr" $f > $O/${tag}_kernel_stats.csv
bash $R/tools/pmc_traffic.sh $tag --no-pcie --steps 10 --warmup 2 "$@" > /dev/null 2>&1
cat $O/${tag}_kernel_stats.csv | cut -c1-160; cat $O/pmc_traffic_$tag.json | tail -n 8
