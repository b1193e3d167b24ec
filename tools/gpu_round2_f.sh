#!/bin/bash
# round 2, f-4: fused ConvLSTM parity + timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_convlstm.py -x -q -m gpu > gpurun_out/f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/f_tests.log
tail -30 gpurun_out/f_tests.log
timeout 600 python tools/convlstm_time.py > gpurun_out/f_time.log 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/f_time.log"):
    if l.startswith("{"):
        r=json.loads(l); print(r["shape"], {k:round(v,4) for k,v in r.items() if k!="shape"})
PY
