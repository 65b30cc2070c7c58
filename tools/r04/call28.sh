#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /sys/fs/cgroup/cpu.stat 2>/dev/null
cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null
cat /proc/self/cgroup | head -3
nproc; cat /proc/loadavg
} | tee $O/r04_cgroup.txt
for t in 64 48 32; do
echo "== $t threads"; python tools/r04/e2e_timeline.py $t 30 2>&1 | grep "^rep" | awk '{print $3}' | tr '\n' ' '; echo
grep -i "thrott" /sys/fs/cgroup/cpu.stat 2>/dev/null | tr '\n' ' '; echo
done 2>&1 | tee $O/r04_e2e_thread_counts.txt
cat /proc/loadavg
