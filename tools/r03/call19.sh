#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
t() { echo "== $*"; env "$@" timeout 600 python -m pytest tests/test_extra_channels.py -q -m gpu -k "kw0 or kw1 or kw2" 2>&1 | grep "rs.alpha), float\|TIGHT, \[\|passed\|failed\|FAILED" | cut -c1-200; }
t JXLHIP_DBG_ALPHA=3
t JXLHIP_DBG_ALPHA=4
t JXLHIP_DBG_ALPHA=0 HIP_LAUNCH_BLOCKING=1
t JXLHIP_DBG_ALPHA=0 JXLHIP_FILTERS=generic
