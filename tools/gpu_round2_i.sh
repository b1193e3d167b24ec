#!/bin/bash
# ESIM parity subset + the ESIM bench workloads
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_fast_noise.py tests/test_hip_properties.py tests/test_postops.py -x -q -m gpu > gpurun_out/i_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/i_tests.log
grep -v Warn gpurun_out/i_tests.log | tail -6
for w in cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_dataset_style cfg2_u8 cfg4_u8_256x41x256x256_sum5 cfg2_noise_free; do
  timeout 600 python bench.py --workload $w --steps 30 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/i_bench_$w.json 2> gpurun_out/i_bench_$w.err
  python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/i_bench_{w}.json").read().strip().splitlines()[-1])
    print(w, "ms", round(d["roofline"]["kernel_ms_avg"],4), "p50", round(d["roofline"]["kernel_ms_p50"],4), "frac", round(d["roofline"]["frac"],3), d.get("parity_check"))
except Exception as e:
    print(w, "ERR", e); print(open(f"gpurun_out/i_bench_{w}.err").read()[-1500:])
PY
done
