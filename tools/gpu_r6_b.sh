#!/bin/bash
# round 6, call B: stream workloads (1 and 2 ranks), the new unet test, rebuilt library regression on the touched kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_b; mkdir -p $O
export TMPDIR=/tmp
for wl in cfg4_stream cfg4_stream_staged; do
  python3 bench.py --gpus 1 --steps 30 --warmup 5 --workload $wl > $O/$wl.n1.out 2> $O/$wl.n1.err; echo "$wl n1 rc=$?"; tail -1 $O/$wl.n1.out
done
timeout 1800 python -m pytest tests/test_frontend.py tests/test_hip_properties.py -m gpu -q -x -k "zero_copy or host_fed" > $O/tests.out 2>&1
echo "tests rc=$?"; grep -v amdgpu.ids $O/tests.out | tail -8
grep -v amdgpu.ids $O/cfg4_stream_staged.n1.err | tail -5
grep -v amdgpu.ids $O/cfg4_stream.n1.err | tail -5
python3 tools/train_two_batches_probe.py > $O/two_batches.out 2>&1; tail -30 $O/two_batches.out | grep -v amdgpu
bash tools/valu_count.sh "train_u8_12x201x128x128_sum5" 2>&1 | grep -v amdgpu | tail -3
