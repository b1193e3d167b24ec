#!/bin/bash
# round 6, call A: teardown tests, RNG statistics, the driver's bench command (stderr must be clean)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_a; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_dataset_events.py tests/test_hip_fast_noise.py -m gpu -q > $O/tests.out 2>&1
echo "tests rc=$?"; grep -v amdgpu.ids $O/tests.out | tail -15
cp gpurun_out/rng_statistics.json $O/ 2>/dev/null
S=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"
tail -1 $O/bench.out
echo "--- stderr (without amdgpu.ids) ---"; grep -v amdgpu.ids $O/bench.err | tail -30
cp bench_extra.json $O/ 2>/dev/null
