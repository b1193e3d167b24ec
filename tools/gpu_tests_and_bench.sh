#!/bin/bash
# Run ON THE GPU BOX: the DRIVER's bench command first (contract line last on stdout + sidecar), then the full -m gpu suite
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/drv_bench.json 2> gpurun_out/drv_bench.err
brc=$?
cp bench_extra.json gpurun_out/drv_bench_extra.json 2>/dev/null
echo "bench rc=$brc wall=$(( $(date +%s) - t0 ))s line_bytes=$(tail -n 1 gpurun_out/drv_bench.json | wc -c) stdout_lines=$(wc -l < gpurun_out/drv_bench.json)"
python - <<'PY'
import json
try:
    d=json.loads(open("gpurun_out/drv_bench.json").read().strip().splitlines()[-1])
    print("headline", d["config"]["workload"], round(d["value"]), "ms/step", round(d["ms_per_step"],4), "kernel ms", round(d["roofline"]["kernel_ms_avg"],4), "frac", round(d["roofline"]["frac"],3), d.get("parity_check"), "cpu", d["cpu_baseline"] and round(d["cpu_baseline"]["value"],2))
    x=json.load(open("gpurun_out/drv_bench_extra.json"))
    print("seconds", x["seconds"])
    for k,v in x.get("also_measured",{}).items():
        print(" ", k, {kk: (round(vv,4) if isinstance(vv,float) else vv) for kk,vv in v.items() if kk in ("kernel_ms_avg","kernel_ms_p50","frac_of_hbm_peak","parity_check","error","integration_levels_samples_per_s")})
except Exception as e:
    print("ERR", e); print(open("gpurun_out/drv_bench.err").read()[-3000:])
PY
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/tb_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/tb_tests.log
grep -v Warning gpurun_out/tb_tests.log | tail -15
