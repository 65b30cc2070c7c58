#!/bin/bash
# round-2 evidence: GPU suite, then per-config bench + rocprof kernel stats + PMC traffic
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r02w_tests.txt 2>&1; tail -3 $O/r02w_tests.txt
for c in c3 c1 c2 c5 c4; do
  bash tools/profile_round.sh r02_$c --config $c > $O/r02w_$c.txt 2>&1
  tail -n 14 $O/r02w_$c.txt | cut -c1-170
done
bash tools/profile_round.sh r02_c3_epf3 --config c3 --epf 3 > $O/r02w_c3_epf3.txt 2>&1
bash tools/profile_round.sh r02_real8k --config c3 --mix real4k > $O/r02w_real8k.txt 2>&1
