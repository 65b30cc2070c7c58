#!/bin/bash
# Device ISA + resource usage of one HIP source (no GPU needed): tools/isa.sh kernels_filters_fast.hip [out.s]
src=$1; out=${2:-/tmp/dis/$(basename $src .hip).s}
mkdir -p $(dirname $out)
cd $(dirname $0)/../libjxl_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -x hip --cuda-device-only -S \
  -Rpass-analysis=kernel-resource-usage $src -o $out 2> /tmp/dis/remarks.txt
python3 - <<'PY'
import re,subprocess
rows=[];cur={}
for l in open('/tmp/dis/remarks.txt'):
    m=re.search(r'remark: (?:\S+ )?\s*(Function Name|VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|SGPRs|VGPRs Spill): (\S+)',l.replace('TotalSGPRs','SGPRs'))
    if 'error' in l: print(l.rstrip())
    if not m: continue
    k,v=m.groups()
    if k=='Function Name':
        if cur: rows.append(cur)
        cur={'name':subprocess.run(['c++filt',v],capture_output=True,text=True).stdout.strip()[:90]}
    else: cur[k]=v
if cur: rows.append(cur)
for r in rows:
    print("%-90s vgpr %4s sgpr %4s scratch %4s occ %s lds %s"%(r['name'],r.get('VGPRs'),r.get('SGPRs'),r.get('ScratchSize [bytes/lane]'),r.get('Occupancy [waves/SIMD]'),r.get('LDS Size [bytes/block]')))
PY
echo "ISA in $out"
