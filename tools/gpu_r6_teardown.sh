#!/bin/bash
# round 6: diagnose + verify the spawned-worker teardown
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_teardown; mkdir -p $O
export TMPDIR=/tmp
echo "== A: hook OFF, terminate backtrace preloaded, 4 rounds" > $O/summary.txt
V2V_WORKER_EXIT_HOOK=0 LD_PRELOAD=$PWD/tools/diag/terminate_trace.so timeout 600 python tools/spawn_teardown_probe.py --rounds 4 > $O/a.out 2> $O/a.err
echo "rc=$? $(cat $O/a.out)" >> $O/summary.txt
grep -c "terminate called\|terminate_trace" $O/a.err >> $O/summary.txt
echo "== B: hook ON, 6 rounds" >> $O/summary.txt
LD_PRELOAD=$PWD/tools/diag/terminate_trace.so timeout 600 python tools/spawn_teardown_probe.py --rounds 6 > $O/b.out 2> $O/b.err
echo "rc=$? $(cat $O/b.out)" >> $O/summary.txt
grep -c "terminate called\|terminate_trace" $O/b.err >> $O/summary.txt
echo "== C: hook ON, non-persistent, 4 rounds" >> $O/summary.txt
timeout 600 python tools/spawn_teardown_probe.py --rounds 4 --persistent 0 > $O/c.out 2> $O/c.err
echo "rc=$? $(cat $O/c.out)" >> $O/summary.txt
grep -c "terminate called" $O/c.err >> $O/summary.txt
echo "== D: tests" >> $O/summary.txt
timeout 1200 python -m pytest tests/test_hip_dataset_events.py -m gpu -x -q -k "spawn" > $O/d.out 2>&1
tail -5 $O/d.out >> $O/summary.txt
cat $O/summary.txt
grep -v "amdgpu.ids" $O/a.err | head -80
