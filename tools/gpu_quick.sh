#!/bin/bash
# quick GPU check: build, smoke, gpu tests, bench a few workloads (run through gpurun from the repo root)
mkdir -p gpurun_out
(python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3) > gpurun_out/smoke.log
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/pytest_gpu.log
for wl in ${WORKLOADS:-cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_noise_on cfg2_u8 train_u8_12x201x128x128_sum5}; do
  (timeout 600 python bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline 2>&1 | tail -1) > gpurun_out/bench_$wl.log
done
cat gpurun_out/smoke.log gpurun_out/pytest_gpu.log
cat gpurun_out/bench_*.log | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l); continue
    r=d['roofline']
    print(d['config']['workload'], round(d['value']), 'grids/s', round(r['kernel_ms_avg'],4), 'ms(avg)', round(r['kernel_ms_p50'],4), 'p50', round(r['achieved']), 'GB/s', round(r['frac'],3), d['parity_check'])
"
