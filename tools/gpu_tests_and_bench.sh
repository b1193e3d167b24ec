#!/bin/bash
# Run ON THE GPU BOX: the full -m gpu suite, then the default bench line with every also_measured workload
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/tb_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/tb_tests.log
tail -15 gpurun_out/tb_tests.log
timeout 900 python bench.py > gpurun_out/tb_bench.json 2> gpurun_out/tb_bench.err
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/tb_bench.json").read().strip().splitlines()[-1])
    print("headline", d["config"]["workload"], d["value"], "ms", d["roofline"]["kernel_ms_avg"], "frac", round(d["roofline"]["frac"],3), d.get("parity_check"))
    for k,v in d.get("also_measured",{}).items():
        print(" ", k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("kernel_ms_avg","kernel_ms_p50","frac_of_hbm_peak","parity_check","error")})
except Exception as e:
    print("ERR", e); print(open("gpurun_out/tb_bench.err").read()[-2000:])
PY
