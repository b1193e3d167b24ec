#!/bin/bash
# round 2, f-4: fused ConvLSTM parity + timing + cfg5 with/without
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_convlstm.py -x -q -m gpu > gpurun_out/f_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/f_tests.log
tail -30 gpurun_out/f_tests.log
timeout 600 python tools/convlstm_time.py > gpurun_out/f_time.log 2>&1
tail -8 gpurun_out/f_time.log
for w in cfg5_pipeline_plus_e2vid_bf16 cfg5_fused_convlstm; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 3 --no-also > gpurun_out/f_bench_$w.json 2> gpurun_out/f_bench_$w.err
  python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/f_bench_{w}.json").read().strip().splitlines()[-1])
    print(w, d["value"], d["ms_per_step"], d["config"].get("launch"))
except Exception as e:
    print(w, "ERR", e); print(open(f"gpurun_out/f_bench_{w}.err").read()[-1500:])
PY
done
