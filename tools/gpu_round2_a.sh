#!/bin/bash
# round-2 GPU check A: ubenches, smoke, gpu tests, noise workloads
mkdir -p gpurun_out
./tools/ubench/rng_rates > gpurun_out/rng_rates.txt 2>&1
./tools/ubench/valu_rates > gpurun_out/valu_rates.txt 2>&1
(python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3) > gpurun_out/smoke.log
(timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/pytest_gpu.log
for wl in cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_noise_on cfg2_noise_on_fast cfg2_u8 cfg3_v2e_f32_256x32x256x256_bilinear5; do
  (timeout 600 python bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_$wl.log
done
cat gpurun_out/rng_rates.txt gpurun_out/smoke.log gpurun_out/pytest_gpu.log
cat gpurun_out/bench_*.log | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l); continue
    r=d['roofline']
    print(d['config']['workload'], round(d['value']), 'grids/s', round(r['kernel_ms_avg'],4), 'ms(avg)', round(r['kernel_ms_p50'],4), 'p50', round(r['achieved']), 'GB/s', round(r['frac'],3), d['parity_check'])
"
