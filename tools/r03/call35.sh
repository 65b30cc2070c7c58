#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 900 python -m pytest tests/test_gpu_vs_reference.py -q -m gpu -k "packed" 2>&1 | grep -v amdgpu.ids | tail -12 | cut -c1-300
timeout 400 python tools/packed_bench.py 2>&1 | grep "8K"
