#!/bin/bash
# round-2 GPU check B: gpu tests + main-lib benches + prebuilt kernel variants
mkdir -p gpurun_out
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/pytest_gpu.log
cat gpurun_out/pytest_gpu.log
fmt='
import sys,json
d=json.loads(sys.stdin.readline()); r=d["roofline"]
print(sys.argv[1], d["config"]["workload"][:14], round(r["kernel_ms_p50"],4), "ms p50", round(r["kernel_ms_avg"],4), "avg", round(r["achieved"]), "GB/s", round(r["frac"],3), d["parity_check"])'
for wl in cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_noise_on cfg2_noise_on_fast cfg2_u8 cfg2_asym cfg4_u8_256x41x256x256_sum5; do
  python bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$fmt" main
done
for lib in gpurun_variants/lib_*.so; do
  for wl in ${VWORKLOADS:-cfg2_noise_on cfg2_esim_f32_256x32x256x256_bilinear5}; do
    V2V_HIP_LIB=$PWD/$lib python bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$fmt" $lib
  done
done
