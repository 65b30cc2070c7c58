#!/bin/bash
# GPU box: the same env_sweep measurements on several builds of the library in ONE session (same box, same clocks):
# usage: tools/r06/ab.sh <out file> <variant.so> ...   ("-" = the product's library)
out=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2; do
for so in "$@"; do
  [ "$so" = "-" ] && unset JXLHIP_SO || export JXLHIP_SO=$R/$so
  echo "## rep $rep library ${so}" >> $out
  for mix in d1 real4k 5:1 18:1 4:1; do
    echo "# mix $mix" >> $out
    python tools/r06/env_sweep.py --mix $mix --steps 40 --reps 3 --envs "" 2>/dev/null | grep '^{' >> $out
  done
done
done
