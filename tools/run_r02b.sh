#!/bin/bash
# GPU box: parity tests + bench of the lean filter kernel
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r02b_pytest.txt 2>&1
tail -5 $O/r02b_pytest.txt
for res in 512 768; do
  echo "== resident $res" >> $O/r02b_bench.txt
  JXLHIP_FILTER_RESIDENT=$res python bench.py --no-cpu-baseline --steps 50 --warmup 5 >> $O/r02b_bench.txt 2>&1
done
for dbg in 4 8 12; do
  echo "== JXLHIP_DEBUG=$dbg" >> $O/r02b_bench.txt
  JXLHIP_DEBUG=$dbg python bench.py --no-cpu-baseline --steps 50 --warmup 5 >> $O/r02b_bench.txt 2>&1
done
echo "== epf2" >> $O/r02b_bench.txt
python bench.py --no-cpu-baseline --steps 30 --warmup 5 --epf 2 >> $O/r02b_bench.txt 2>&1
echo "== filters off" >> $O/r02b_bench.txt
python bench.py --no-cpu-baseline --steps 30 --warmup 5 --epf 0 --gab 0 >> $O/r02b_bench.txt 2>&1
grep -o '"value": [0-9.]*\|kernel_ms": {[^}]*}\|^==.*' $O/r02b_bench.txt
