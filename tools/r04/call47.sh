#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
JXLHIP_FUSE=1 bash tools/profile_round.sh r04_c3_epf3_fused --epf 3 2>&1 | tail -14
cd $R; timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 | tee $O/r04_final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
