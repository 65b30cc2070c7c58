#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for e in 0 1 2; do
echo "== JXLHIP_DBG_ALPHA=$e"
JXLHIP_DBG_ALPHA=$e timeout 600 python -m pytest tests/test_extra_channels.py -q -m gpu -k "kw0 or kw1 or kw2" 2>&1 | grep -v "^E    *+\|amdgpu.ids" | grep "AssertionError\|passed\|failed\|FAILED\|assert" | cut -c1-300
done
