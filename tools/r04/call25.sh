#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
grep MHz /proc/cpuinfo | sort | uniq -c | sort -rn | head -3
cat /sys/devices/system/cpu/cpu0/cpufreq/scaling_governor 2>/dev/null
python tools/r04/wp_variants.py 2>&1 | tee $O/r04_wp_variants_box.txt
grep MHz /proc/cpuinfo | sort -k4 -n | tail -2
