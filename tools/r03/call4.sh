#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_vs_reference.py -m gpu -q --tb=short 2>&1 | tail -8
echo "== packed, default path choice"; timeout 300 python tools/packed_bench.py
echo "== packed, fused (PC) forced"; JXLHIP_FUSE=1 timeout 300 python tools/packed_bench.py
echo "== packed, fused (single-wave) forced"; JXLHIP_FUSE=1 JXLHIP_FUSED_PC=0 timeout 300 python tools/packed_bench.py
