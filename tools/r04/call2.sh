#!/bin/bash
# round 4, GPU call 2: parity of the tile producer (opt-in) and of the fused / stripe tests after the k_prepare changes
# (every cell writes its cell_info, alternating counter blocks); the c3 step and its launch timeline
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "fused or tile or stripes or c1_ or mixed" 2>&1 | tail -25 > $O/r04_call2_tests.txt
cat $O/r04_call2_tests.txt
q() { bash tools/quick.sh "$1" --no-pcie --steps 200 --warmup 20 "${@:2}"; }
{
for rep in 1 2; do q ""; done
q "" --config c2; q "" --config c1; q "" --mix real4k; q "" --config c4; q "" --width 3840 --height 2160
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python $R/bench.py --no-cpu-baseline --no-pcie --steps 60 --warmup 5 > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -1); echo "trace: $f"; tail -3 /tmp/tl.log; python $R/tools/timeline.py $f
} 2>&1 | tee $O/r04_call2_bench.txt
