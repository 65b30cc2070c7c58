#!/bin/bash
# Runs on the GPU box: HBM traffic per kernel launch from the TCC counters, one
# counter per pass (MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE cannot
# share a pass; FETCH_SIZE under-counts wide streaming reads 2x on gfx950 -- the
# run also times a known-size copy to calibrate both).
tag=${1:-run}; shift
export JXLHIP_BENCH_NO_GRAPH=1  # (profiling / experiment runs: no hipGraph side measurement)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$ctr
  timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$ctr -- \
    python $R/bench.py --no-cpu-baseline --frames-in-flight 1 --calib-copy "$@" > $O/pmc_${ctr}_$tag.log 2>&1
  f=$(find /tmp/pmc_$ctr -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp $f $O/pmc_${ctr}_$tag.csv
done
python $R/tools/pmc_summarize.py $O/pmc_FETCH_SIZE_$tag.csv $O/pmc_WRITE_SIZE_$tag.csv > $O/pmc_traffic_$tag.json
cat $O/pmc_traffic_$tag.json
