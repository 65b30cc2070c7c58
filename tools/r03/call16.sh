#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
timeout 900 python -m pytest tests/test_extra_channels.py tests/test_djxl.py tests/test_codestream.py tests/test_seam.py -q -x -m gpu 2>&1 | tail -15
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "mfma or matrix_cores" 2>&1 | tail -4
timeout 200 python tools/packed_bench.py 2>&1 | tail -8
