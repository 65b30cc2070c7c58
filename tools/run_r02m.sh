#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "epf or pipeline or stripes" > $O/r02m_tests.txt 2>&1; tail -5 $O/r02m_tests.txt
bash tools/kstats.sh "A=1" --config c3 --epf 3 --no-pcie --steps 20 --warmup 5 > $O/r02m_c3_epf3.txt 2>&1; cat $O/r02m_c3_epf3.txt
bash tools/kstats.sh "JXLHIP_FUSE=0" --config c3 --epf 2 --no-pcie --steps 20 --warmup 5 > $O/r02m_c3_epf2_two.txt 2>&1; cat $O/r02m_c3_epf2_two.txt
bash tools/kstats.sh "JXLHIP_FUSE=0" --config c3 --epf 1 --no-pcie --steps 20 --warmup 5 > $O/r02m_c3_epf1_two.txt 2>&1; cat $O/r02m_c3_epf1_two.txt
bash tools/kstats.sh "A=1" --config c3 --epf 2 --no-pcie --steps 20 --warmup 5 > $O/r02m_c3_epf2.txt 2>&1; cat $O/r02m_c3_epf2.txt
