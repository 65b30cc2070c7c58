#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 1200 python -m pytest tests/test_extra_channels.py tests/test_djxl.py tests/test_seam.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15 | cut -c1-300
