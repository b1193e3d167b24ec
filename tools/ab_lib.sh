#!/bin/bash
# Run ON THE GPU BOX: same-box A/B of two builds of the library over bench.py's headline + secondary kernel workloads, interleaved.
# usage: ab_lib.sh <variant .so> [rounds]     (A = v2v_amd/libv2v_hip.so, B = the variant)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
B=$1; R=${2:-2}
for r in $(seq 1 $R); do
  for tag in A B; do
    if [ $tag = A ]; then unset V2V_HIP_LIB; else export V2V_HIP_LIB=$B; fi
    python bench.py --steps 50 --warmup 10 --no-cpu-baseline --kernels-only --full --extra-out gpurun_out/ab_${tag}_$r.json > gpurun_out/ab_${tag}_$r.line 2>/dev/null
  done
done
unset V2V_HIP_LIB
python - "$R" <<'PY'
import json, sys
R = int(sys.argv[1])
rows = {}
for tag in "AB":
    for r in range(1, R + 1):
        d = json.load(open(f"gpurun_out/ab_{tag}_{r}.json"))
        rows.setdefault("headline", {}).setdefault(tag, []).append(d["roofline"]["kernel_ms_p50"])
        for k, v in d.get("also_measured", {}).items():
            if "kernel_ms_p50" in v:
                rows.setdefault(k, {}).setdefault(tag, []).append(v["kernel_ms_p50"])
for k, v in rows.items():
    a, b = min(v["A"]), min(v["B"])
    print(f"{k:48s} A {a:8.4f}  B {b:8.4f}  B/A {b / a:6.3f}   A runs {[round(x, 4) for x in v['A']]} B runs {[round(x, 4) for x in v['B']]}")
PY
