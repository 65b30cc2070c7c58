#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 400 python tools/packed_bench.py 2>&1 | grep "8K"
