#!/bin/bash
# Run ON THE GPU BOX: where the waves of the dominant kernel spend their cycles -- one rocprofv3 --pmc pass (8 SQ counters) per workload:
# SQ_WAVE_CYCLES = SQ_ACTIVE_INST_ANY (issuing) + SQ_WAIT_INST_ANY (ready but not issued: pipe busy / dependency) + SQ_WAIT_ANY (parked at
# s_waitcnt or a barrier), and the active share by pipe (vector ALU, scalar, LDS, vector memory).  Units: quad-cycles summed over waves.
# usage: stall_breakdown.sh "<workloads>" [out.json]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=${2:-$REPO/gpurun_out/stall_breakdown.json}
echo "{" > $OUT.tmp
first=1
for wl in ${1:-cfg2_esim_f32_256x32x256x256_bilinear5}; do
  rm -rf /tmp/sb_$wl
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/sb_$wl -o sb -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --workload $wl > /tmp/sb_$wl.json 2>/dev/null
  rm -rf /tmp/sb2_$wl
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/sb2_$wl -o sb -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-also --workload $wl > /dev/null 2>&1
  [ $first = 1 ] || echo "," >> $OUT.tmp
  first=0
  python3 - "$wl" /tmp/sb_$wl /tmp/sb2_$wl /tmp/sb_$wl.json >> $OUT.tmp << 'PY'
import csv, glob, json, sys
wl, d1, d2, jf = sys.argv[1:5]
acc = {}
for d in (d1, d2):
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if ("voxel_kernel" in k and "shot_sum" not in k) or "frontend_tile" in k:
                acc.setdefault((k.split("(")[0][-90:], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
out = {}
for (k, c), v in acc.items():
    out.setdefault(k, {})[c] = sum(v) / len(v)
try:
    ms = json.loads(open(jf).read().strip().splitlines()[-1])["roofline"]["kernel_ms_avg"]
except Exception:
    ms = None
res = {"kernel_ms_under_profiler": ms, "kernels": {}}
for k, m in out.items():
    wc = max(m.get("SQ_WAVE_CYCLES", 0), 1.0)
    res["kernels"][k] = {"counters_per_launch": m, "share_of_wave_cycles": {
        "issuing (ACTIVE_INST_ANY)": round(m.get("SQ_ACTIVE_INST_ANY", 0) / wc, 3), "ready, not issued (WAIT_INST_ANY)": round(m.get("SQ_WAIT_INST_ANY", 0) / wc, 3),
        "  of which LDS issue (WAIT_INST_LDS)": round(m.get("SQ_WAIT_INST_LDS", 0) / wc, 3), "parked at waitcnt / barrier (WAIT_ANY)": round(m.get("SQ_WAIT_ANY", 0) / wc, 3),
        "active: vector ALU": round(m.get("SQ_ACTIVE_INST_VALU", 0) / wc, 3), "active: scalar": round(m.get("SQ_ACTIVE_INST_SCA", 0) / wc, 3),
        "active: LDS": round(m.get("SQ_ACTIVE_INST_LDS", 0) / wc, 3), "active: vector memory": round(m.get("SQ_ACTIVE_INST_VMEM", 0) / wc, 3)},
        "instructions_per_wave": {c[9:].lower(): round(m[c] / max(m.get("SQ_WAVES", 1), 1), 1) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_INSTS_SMEM") if c in m}}
print(json.dumps(wl) + ": " + json.dumps(res, indent=1))
PY
done
echo "}" >> $OUT.tmp
mv $OUT.tmp $OUT
python3 -c "
import json,sys
d=json.load(open('$OUT'))
for wl,r in d.items():
    for k,v in r['kernels'].items():
        print(wl[:30], k[-60:], r['kernel_ms_under_profiler']); print('   ', v['share_of_wave_cycles']); print('   ', v['instructions_per_wave'])
"
