#!/bin/bash
mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_yolo_parity.py -m gpu -q -rf --tb=short -k "tight" 2>&1 | tail -12
python - <<'PY'
import json
j=json.load(open('gpurun_out/parity_report.json'))
for k,v in j.items(): print(k, {a:(round(b,6) if isinstance(b,float) else b) for a,b in v.items()})
PY
