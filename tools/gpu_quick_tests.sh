#!/bin/bash
# Run ON THE GPU BOX: the whole -m gpu suite (no -x), tail of the log
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/tests_gpu.log 2>&1
echo "tests rc=$?" >> gpurun_out/tests_gpu.log
grep -v Warning gpurun_out/tests_gpu.log | tail -12
