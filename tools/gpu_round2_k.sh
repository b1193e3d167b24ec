#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/k_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/k_tests.log
grep -v "Warn\|pin_memory" gpurun_out/k_tests.log | tail -4
bash tools/ab_variants.sh "cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_dataset_style cfg3_v2e_f32_256x32x256x256_bilinear5 cfg2_u8" 2
