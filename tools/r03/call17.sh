#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for e in 1 0; do
echo "== JXLHIP_SPARSE_UPLOAD=$e"
JXLHIP_SPARSE_UPLOAD=$e timeout 600 python -m pytest tests/test_extra_channels.py -q -m gpu 2>&1 | grep -v "^E    *+\|amdgpu.ids" | grep "AssertionError\|passed\|failed\|FAILED" | cut -c1-600
done
