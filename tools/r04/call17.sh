#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
bash tools/regen_profiles.sh r04 74c77f483ffa
