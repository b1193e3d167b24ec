#!/bin/bash
# Run ON THE GPU BOX: A/B the main library against every prebuilt variant in gpurun_variants/ on the same box, interleaved.
# usage: ab_variants.sh "<workloads>" [rounds]
cd "$GRAFT_REPO_ROOT" || exit 1
WLS=${1:-cfg2_esim_f32_256x32x256x256_bilinear5}
ROUNDS=${2:-2}
for r in $(seq 1 $ROUNDS); do
  for wl in $WLS; do
    for lib in main gpurun_variants/lib_*.so; do
      if [ "$lib" = main ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$PWD/$lib; fi
      python bench.py --steps 40 --warmup 5 --workload $wl --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$r', '$lib'.split('/')[-1][:22].ljust(22), d['config']['workload'][:22].ljust(22), 'p50', round(r['kernel_ms_p50'],4), 'avg', round(r['kernel_ms_avg'],4), d['parity_check'])"
    done
  done
done
