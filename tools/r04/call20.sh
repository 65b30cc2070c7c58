#!/bin/bash
# e2e after LPT order of the AC groups + dense-first for heavy sections
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
python - <<'PY' 2>&1 | tee $O/r04_call20_e2e.txt
import json, torch, bench
print(json.dumps(bench.e2e_block(torch, 0), indent=1))
PY
timeout 900 python -m pytest tests -q -m gpu -k "codestream or djxl or extra or conformance or frame_header or alpha" 2>&1 | tail -5 | tee $O/r04_call20_tests.txt
