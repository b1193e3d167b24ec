#!/bin/bash
# round 2: v2e pre-pass rewrite -- parity + timing
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_v2e.py tests/test_hip_fullsize.py tests/test_hip_properties.py -x -q -m gpu > gpurun_out/g_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/g_tests.log
tail -15 gpurun_out/g_tests.log
for w in cfg3_v2e_f32_256x32x256x256_bilinear5 cfg3_v2e_u8; do
  timeout 600 python bench.py --workload $w --steps 20 --warmup 3 --no-also --no-cpu-baseline > gpurun_out/g_bench_$w.json 2> gpurun_out/g_bench_$w.err
  python - "$w" <<'PY'
import json,sys
w=sys.argv[1]
try:
    d=json.loads(open(f"gpurun_out/g_bench_{w}.json").read().strip().splitlines()[-1])
    print(w, "ms", d["roofline"]["kernel_ms_avg"], "p50", d["roofline"]["kernel_ms_p50"], d.get("parity_check"))
except Exception as e:
    print(w, "ERR", e); print(open(f"gpurun_out/g_bench_{w}.err").read()[-1500:])
PY
done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g_stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3_v2e_f32_256x32x256x256_bilinear5 --steps 10 --warmup 2 --no-also --no-cpu-baseline > /dev/null 2>&1
head -4 $(find /tmp/g_stats -name "*kernel_stats.csv" | head -1) | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g_stats8 -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload cfg3_v2e_u8 --steps 10 --warmup 2 --no-also --no-cpu-baseline > /dev/null 2>&1
head -4 $(find /tmp/g_stats8 -name "*kernel_stats.csv" | head -1) | cut -c1-200
