#!/bin/bash
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12) > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
fmt='
import sys,json
d=json.loads(sys.stdin.readline()); r=d["roofline"]
print(sys.argv[1], d["config"]["workload"][:14], round(r["kernel_ms_p50"],4), "ms p50", round(r["kernel_ms_avg"],4), "avg", round(r["achieved"]), "GB/s", round(r["frac"],3), d["parity_check"])'
for wl in ${WORKLOADS:-cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_noise_on cfg3_v2e_f32_256x32x256x256_bilinear5 cfg3_v2e_u8}; do
  python bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$fmt" main
done
bash tools/valu_count.sh "cfg3_v2e_f32_256x32x256x256_bilinear5"
