#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 600 python -m pytest tests/test_djxl.py -q -m gpu -k "kw5 or kw7" 2>&1 | grep -v amdgpu.ids | grep "declines\|passed\|failed\|seam: " | cut -c1-400 | head -12
