#!/bin/bash
# Run ON THE GPU BOX: dynamic vector-instruction mix of the dominant kernel by class (two rocprofv3 --pmc passes), per wave and time step.
# usage: valu_mix.sh "<workloads>" [out.json]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
OUT=${2:-$REPO/gpurun_out/valu_mix.json}
echo "{" > $OUT.tmp
first=1
for wl in ${1:-cfg2_esim_f32_256x32x256x256_bilinear5}; do
  rm -rf /tmp/vm1_$wl /tmp/vm2_$wl
  ARGS="$REPO/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-also --workload $wl"
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT64 --output-format csv -d /tmp/vm1_$wl -o p -- python3 $ARGS > /tmp/vm_$wl.json 2>/dev/null
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH --output-format csv -d /tmp/vm2_$wl -o p -- python3 $ARGS > /dev/null 2>&1
  [ $first = 1 ] || echo "," >> $OUT.tmp
  first=0
  python3 - "$wl" /tmp/vm1_$wl /tmp/vm2_$wl /tmp/vm_$wl.json >> $OUT.tmp << 'PY'
import csv, glob, json, sys
wl, d1, d2, jf = sys.argv[1:5]
acc = {}
for d in (d1, d2):
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            if "voxel_kernel" in k and "shot_sum" not in k:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
m = {c: sum(v) / len(v) for c, v in acc.items()}
try:
    steps = json.loads(open(jf).read().strip().splitlines()[-1])["config"]["frames"] - 1
except Exception:
    steps = 1
w = max(m.get("SQ_WAVES", 1), 1)
per = {c[9:].lower(): round(v / w / steps, 2) for c, v in m.items() if c != "SQ_WAVES"}
known = sum(per.get(k, 0) for k in ("valu_add_f64", "valu_mul_f64", "valu_fma_f64", "valu_trans_f64", "valu_cvt", "valu_int64", "valu_int32", "valu_add_f32", "valu_mul_f32", "valu_fma_f32", "valu_trans_f32"))
per["valu_unclassified"] = round(per.get("valu", 0) - known, 2)
print(json.dumps(wl) + ": " + json.dumps({"per_wave_and_time_step": per, "time_steps": steps}, indent=1))
PY
done
echo "}" >> $OUT.tmp
mv $OUT.tmp $OUT
cat $OUT
