#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of library builds under the DRIVER's protocol (a cold process, 5 warm-ups, 20 timed launches: mostly
# clock ramp, where a vector-issue-bound kernel is slower than once settled).  usage: ab_driver_protocol.sh <rounds> <libA|-> <libB> [...]
cd "$GRAFT_REPO_ROOT" || exit 1
R=$1; shift
for r in $(seq 1 $R); do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$lib; fi
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('%-36s ms/step %.4f  kernel avg %.4f  p50 %.4f  settled %s' % ('$lib', d['ms_per_step'], r['kernel_ms_avg'], r['kernel_ms_p50'], r.get('kernel_ms_settled_p50')))"
  done
done
