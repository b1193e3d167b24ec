#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_e; mkdir -p $O
export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_hip_dataset_events.py tests/test_hip_properties.py tests/test_frontend.py -m gpu -q > $O/tests.out 2>&1
echo "tests rc=$?"; grep -v amdgpu.ids $O/tests.out | tail -6
bash tools/profile_driver_cmd.sh 2>&1 | grep -v amdgpu.ids | tail -3
grep -c "Aborted\|terminate called" gpurun_out/prof_driver_cmd/err.log
