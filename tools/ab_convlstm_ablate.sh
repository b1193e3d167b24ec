#!/bin/bash
# Run ON THE GPU BOX: tools/cl_tile_probe.py (ms per ConvLSTM step by tile) for the main library and the timing-ablation builds in
# gpurun_variants/ (-DV2V_CL_ABLATE_STAGE / _READS / _MFMA: results invalid, time only), interleaved on one box.
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2; do
  for lib in main gpurun_variants/lib_*.so; do
    if [ "$lib" = main ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$PWD/$lib; fi
    echo "== $r $lib"; python tools/cl_tile_probe.py 2>/dev/null | head -3
  done
done
