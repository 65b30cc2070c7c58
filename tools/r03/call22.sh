#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
JXLHIP_DBG_SP=1 timeout 600 python -m pytest tests/test_extra_channels.py -q -s -m gpu -k "kw2" 2>&1 | grep "DBG_SP\|passed\|failed\|FAILED" | cut -c1-250
echo ==== with kw0 kw1 before
JXLHIP_DBG_SP=1 timeout 600 python -m pytest tests/test_extra_channels.py -q -s -m gpu -k "kw0 or kw1 or kw2" 2>&1 | grep "DBG_SP\|passed\|failed\|FAILED" | cut -c1-250
