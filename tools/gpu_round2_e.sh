#!/bin/bash
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
(time python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err) 2>&1 | tail -3
tail -3 gpurun_out/bench_default.err
python - << 'PY'
import json
d = json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("HEADLINE", d["config"]["workload"], d["config"]["sim_params"], round(d["value"]), "grids/s", round(r["kernel_ms_avg"], 4), "ms", round(r["frac"], 3), d["parity_check"], "backend", d["dist_backend"])
print("cpu", d["cpu_baseline"]["value"] if d["cpu_baseline"] else None, (d["cpu_baseline_numpy_pool"] or {}).get("value"), (d["cpu_baseline_c_omp"] or {}).get("value"))
for k, v in (d["also_measured"] or {}).items():
    print(f"  {k:42s}", (round(v["kernel_ms_avg"], 4), round(v["frac_of_hbm_peak"], 3), round(v["grids_per_s"]), v["parity_check"]) if "error" not in v else v)
print("host_input", d["host_input"])
PY
