#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 400 python -m pytest tests/test_gpu_vs_reference.py tests/test_real_streams.py tests/test_codestream.py tests/test_seam.py tests/test_djxl.py -m gpu -q --tb=short --timeout 100 -x 2>&1 | tail -4
for sp in 1 0; do echo "== JXLHIP_SPARSE_UPLOAD=$sp, 8K genuine stream"; JXLHIP_SPARSE_UPLOAD=$sp timeout 120 python tools/e2e_real.py oracle/_ref/real_8k_d1.npz 2>&1 | grep "^runner\|kernels alone\|Error\|error" ; done
echo "== 4K"; timeout 120 python tools/e2e_real.py 2>&1 | grep "^runner"
