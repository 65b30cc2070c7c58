#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
t() { echo "== $*"; env "$@" timeout 600 python -m pytest tests/test_extra_channels.py -q -m gpu -k "kw0 or kw1 or kw2" 2>&1 | grep "TIGHT, \[\|passed\|failed\|FAILED" | cut -c1-200; }
t JXLHIP_DBG_ALPHA=0
t HSA_NO_SCRATCH_RECLAIM=1
t JXLHIP_DBG_WARM=1
t JXLHIP_DBG_WARM=1
