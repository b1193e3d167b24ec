#!/bin/bash
# EXPERIMENT driver (GPU box) for docs/experiments/convlstm_pingpong_schedule.patch: race screen (bit-identity with the shipped
# 256-pixel tile over many launches), timings of every tile, and -- with gpurun_variants/lib_*timing.so built with
# -DV2V_CL_TIMING -- the per-segment cycle breakdown.  usage: pp_run.sh [rounds]
cd $GRAFT_REPO_ROOT
for lib in main gpurun_variants/lib_*.so; do
  case $lib in *timing*) continue;; esac
  if [ $lib = main ]; then unset V2V_HIP_LIB; else [ -e $lib ] || continue; export V2V_HIP_LIB=$PWD/$lib; fi
  echo "== $lib"; timeout 300 python tools/pp_screen.py 257 256 ${1:-12} 2>&1 | grep -E "RESULT|MISMATCH" | head -5
  python tools/convlstm_time.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l); print(r['shape'], ' '.join(k[16:-3] + '=' + str(round(v, 4)) for k, v in r.items() if k.startswith('fused_step_only_t')))"
done
for f in gpurun_variants/lib_*timing.so; do [ -e $f ] && V2V_HIP_LIB=$PWD/$f python tools/convlstm_phase_probe.py 2>&1 | grep "tile 25"; done
