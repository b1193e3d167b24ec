#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of library builds on ONE bench workload, interleaved.  usage: ab_workload.sh <workload> <rounds> <libA|-> <libB> [libC ...]
cd "$GRAFT_REPO_ROOT" || exit 1
W=$1; R=$2; shift 2
for r in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$lib; fi
    python bench.py --workload $W --steps 300 --warmup 50 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-36s %s  p10 %.4f  p50 %.4f  avg %.4f ms' % ('$lib', d['config']['workload'], r['kernel_ms_p10'], r['kernel_ms_p50'], r['kernel_ms_avg']))"
  done
done
