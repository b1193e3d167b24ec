#!/bin/bash
# round 6, call C: PREMUL (float64 pre-multiplied Gaussian table, 512-thread workgroups) -- parity, then same-box A/B against the round-5 form
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_c; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_hip_parity.py tests/test_hip_properties.py tests/test_hip_fullsize.py tests/test_hip_fast_noise.py tests/test_loader.py tests/test_postops.py -m gpu -q -x > $O/tests.out 2>&1
echo "tests rc=$?"; grep -v amdgpu.ids $O/tests.out | tail -6
for wl in cfg2_esim_f32_256x32x256x256_bilinear5 cfg2_dataset_style cfg2_u8 cfg4_u8_256x41x256x256_sum5; do
  bash tools/ab_workload.sh $wl 2 - $PWD/gpurun_variants/lib_nopremul.so 2>&1 | grep -v amdgpu.ids
done | tee $O/ab.txt
