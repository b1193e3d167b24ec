#!/bin/bash
# Run ON THE GPU BOX: tools/frontend_time.py for the main library and every gpurun_variants/lib_*.so, interleaved on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
for r in $(seq 1 ${1:-2}); do
  for lib in main gpurun_variants/lib_*.so; do
    if [ "$lib" = main ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$PWD/$lib; fi
    python tools/frontend_time.py 2>&1 | grep "want_imgs=False" | sed "s|^|$r |"
  done
done
