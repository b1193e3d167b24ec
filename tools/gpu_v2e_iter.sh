#!/bin/bash
# Run ON THE GPU BOX: the v2e parity tests + the config-3 bench lines (quick iteration loop of the round-3 v2e work)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_v2e.py tests/test_hip_fuzz.py "tests/test_hip_fullsize.py::test_cfg3_v2e_full_batch_256x32x256x256" ${EXTRA_TESTS} -x -q -m gpu > gpurun_out/v2e_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/v2e_tests.log
tail -6 gpurun_out/v2e_tests.log
for wl in cfg3_v2e_f32_256x32x256x256_bilinear5 cfg3_v2e_u8 ${EXTRA_WL}; do
  timeout 600 python bench.py --workload $wl --no-cpu-baseline --steps 30 --warmup 5 > gpurun_out/v2e_$wl.json 2> gpurun_out/v2e_$wl.err
  python - "$wl" <<'PY'
import json, sys
wl = sys.argv[1]
try:
    d = json.loads(open(f"gpurun_out/v2e_{wl}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(wl, "ms avg %.4f p50 %.4f frac %.3f" % (r["kernel_ms_avg"], r["kernel_ms_p50"], r["frac"]), d.get("parity_check"))
except Exception as e:
    print(wl, "ERR", e); print(open(f"gpurun_out/v2e_{wl}.err").read()[-1500:])
PY
done
