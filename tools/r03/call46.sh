#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for e in "JXLHIP_FUSE=1" "JXLHIP_FUSE=0" "JXLHIP_FUSE=1 JXLHIP_FUSED_PC=0"; do
  bash tools/quick.sh "$e" --config c3 --mix real4k --steps 40
done
for e in "JXLHIP_FUSE=1" "JXLHIP_FUSE=0"; do
  bash tools/quick.sh "$e" --config c2 --mix real4k --steps 60
done
