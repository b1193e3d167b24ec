#!/bin/bash
# round 6: the driver's round-end sequence on one box -- full GPU test suite, smoke, the bench command (plain, then under rocprofv3)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_full; mkdir -p $O
export TMPDIR=/tmp
S=$(date +%s)
timeout 3000 python -m pytest tests -m gpu -q -x > $O/tests.out 2>&1
echo "tests rc=$? in $(( $(date +%s) - S )) s"; grep -v amdgpu.ids $O/tests.out | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
S=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.out 2> $O/bench.err
echo "bench rc=$? wall=$(( $(date +%s) - S ))s"; tail -1 $O/bench.out | cut -c1-600
echo "--- bench stderr (without amdgpu.ids) ---"; grep -v amdgpu.ids $O/bench.err | tail -10
cp bench_extra.json $O/bench_extra.json
bash tools/profile_driver_cmd.sh 2>&1 | grep -v amdgpu.ids | tail -4
