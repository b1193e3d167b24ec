#!/bin/bash
# bench every prebuilt kernel variant in gpurun_variants/ (built with make OUT=... EXTRA=...)
for lib in gpurun_variants/lib_*.so; do
  for wl in ${WORKLOADS:-cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_u8 cfg2_noise_on}; do
    V2V_HIP_LIB=$PWD/$lib python bench.py --steps 30 --warmup 5 --workload $wl --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$lib', d['config']['workload'][:12], round(r['kernel_ms_p50'],4), 'ms p50', round(r['kernel_ms_avg'],4), 'avg', round(r['achieved']), 'GB/s', d['parity_check'])"
  done
done
