#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
export TMPDIR=/tmp
rm -rf gpurun_out/alpha_trace; mkdir -p gpurun_out/alpha_trace
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/alpha_trace -- python -m pytest tests/test_extra_channels.py -q -m gpu -k "kw0 or kw1 or kw2" 2>&1 | grep "TIGHT, \[\|passed\|failed\|FAILED" | cut -c1-200
find gpurun_out/alpha_trace -name "*.csv" | head; du -sh gpurun_out/alpha_trace
