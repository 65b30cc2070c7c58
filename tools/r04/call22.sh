#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
lscpu | grep -i "model name\|^CPU(s)\|thread\|core\|socket\|numa\|MHz" | tee $O/r04_lscpu.txt
cat /sys/devices/system/cpu/cpu0/topology/thread_siblings_list /sys/devices/system/cpu/cpu1/topology/thread_siblings_list 2>/dev/null | tee -a $O/r04_lscpu.txt
python -c "import os; print('affinity', len(os.sched_getaffinity(0)), sorted(os.sched_getaffinity(0))[:8], '...')" | tee -a $O/r04_lscpu.txt
echo "== 64 threads"; JXLHIP_CODESTREAM_VERBOSE=1 python tools/r04/e2e_timeline.py 64 14 2>&1 | grep -v "DC phase" | tee $O/r04_e2e_timeline2.txt
echo "== 64 threads, process confined to logical CPUs 0-63"; taskset -c 0-63 python tools/r04/e2e_timeline.py 64 14 2>&1 | tee $O/r04_e2e_taskset.txt
echo "== 32 threads on 0-31"; taskset -c 0-31 python tools/r04/e2e_timeline.py 32 10 2>&1 | tee $O/r04_e2e_taskset32.txt
timeout 600 python -m pytest tests -q -m gpu -k "codestream or djxl" 2>&1 | tail -3
