#!/bin/bash
# GPU box: bench.py one-liners under different env settings.  usage: tools/quick.sh "ENV=.. ENV=.." [bench args]
envs="$1"; shift
export JXLHIP_BENCH_NO_GRAPH=1  # (profiling / experiment runs: no hipGraph side measurement)
R=${GRAFT_REPO_ROOT:-$PWD}
echo "== env=[$envs] args=[$@]"
env $envs timeout 300 python $R/bench.py --no-cpu-baseline "$@" 2>&1 | grep -o "\"value\": [0-9.]*\|kernel_ms.: {[^}]*}\|Error.*\|error.*" | tr '\n' ' '; echo
