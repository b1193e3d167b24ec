#!/bin/bash
# Run ON THE GPU BOX: A/B the fused ConvLSTM step of the main library against every prebuilt variant in gpurun_variants/
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_convlstm.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do
  for lib in main gpurun_variants/lib_*.so; do
    if [ "$lib" = main ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$PWD/$lib; fi
    python tools/convlstm_time.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        r = json.loads(l)
        print('$r', '$lib'.split('/')[-1][:20].ljust(20), r['shape'], ' '.join(k[16:-3] + '=' + str(round(v, 4)) for k, v in r.items() if k.startswith('fused_step_only_t')), 'TF', round(r['fused_step_tflops']))"
  done
done
