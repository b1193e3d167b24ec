#!/bin/bash
# round 6, call D: config 3 by counters: per kernel (pre-pass / main) VALU instructions, wait cycles, busy fractions
cd "$GRAFT_REPO_ROOT" || exit 1
O=$PWD/gpurun_out/r6_d; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "MemUnitBusy|VALUBusy|MemUnitStalled|LDSBankConflict|VALUUtilization|FetchSize|WriteUnitStalled" | head -20 > $O/avail.txt
for WL in cfg3_v2e_f32_256x32x256x256_bilinear5 cfg3_v2e_u8; do
  ARGS="$GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-also --workload $WL"
  rm -rf /tmp/c3a /tmp/c3b /tmp/c3c
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d /tmp/c3a -o a -- python3 $ARGS > $O/$WL.a.json 2>/dev/null
  rocprofv3 --pmc MemUnitBusy VALUBusy --output-format csv -d /tmp/c3b -o b -- python3 $ARGS > $O/$WL.b.json 2>$O/$WL.b.err
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/c3c -o c -- python3 $ARGS > $O/$WL.c.json 2>$O/$WL.c.err
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3d -o d -- python3 $ARGS > $O/$WL.d.json 2>/dev/null
  python3 - $WL /tmp/c3a /tmp/c3b /tmp/c3c /tmp/c3d > $O/$WL.table.json <<'PY'
import csv, glob, json, sys
wl = sys.argv[1]
acc = {}
for d in sys.argv[2:5]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            k = "pre_pass" if "shot_sum" in r["Kernel_Name"] else "main" if "v2e_voxel_kernel" in r["Kernel_Name"] else None
            if k:
                acc.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {"workload": wl}
for k, c in acc.items():
    out[k] = {n: sum(v) / len(v) for n, v in c.items()}
for path in glob.glob(sys.argv[5] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = "pre_pass" if "shot_sum" in r["Name"] else "main" if "v2e_voxel_kernel" in r["Name"] else None
        if k:
            out.setdefault(k, {})["avg_us"] = float(r["AverageNs"]) / 1e3
print(json.dumps(out, indent=1))
PY
  cat $O/$WL.table.json
done
cat $O/avail.txt
