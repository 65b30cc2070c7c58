#!/bin/bash
# usage: tools/r06/ab2.sh <out> "<mixes>" <variant.so> ...
out=$1; mixes=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2 3; do
for so in "$@"; do
  [ "$so" = "-" ] && unset JXLHIP_SO || export JXLHIP_SO=$R/$so
  for mix in $mixes; do
    echo -n "rep $rep $so mix $mix " >> $out
    python tools/r06/env_sweep.py --mix $mix --steps 40 --reps 3 --no-used-acs --envs "" 2>/dev/null | grep '^{' >> $out
  done
done
done
