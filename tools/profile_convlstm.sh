#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 evidence for the fused ConvLSTM step (tools/convlstm_time.py runs every E2VID encoder shape).
# (1) --kernel-trace --stats, (2) one SQ --pmc pass (matrix-core busy cycles, LDS conflicts).  usage: profile_convlstm.sh <tag>
set -u
TAG=${1:-r02b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG/convlstm
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $REPO/tools/convlstm_time.py > $OUT/time_under_stats.log 2> $OUT/stats.err
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o pmc -- python3 $REPO/tools/convlstm_time.py > $OUT/time_under_pmc.log 2> $OUT/pmc.err
cd $REPO
cp $(find $OUT/stats -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python3 - $OUT <<'PY'
import csv, glob, json, sys
out = sys.argv[1]
acc = {}
for path in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "convlstm_step_kernel" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-40:], r["Grid_Size"], r["Counter_Name"])
            acc.setdefault(key, []).append(float(r["Counter_Value"]))
summ = {}
for (k, g, c), v in acc.items():
    summ.setdefault(f"{k} grid={g}", {})[c] = sum(v) / len(v)
for k, d in summ.items():
    if d.get("SQ_BUSY_CYCLES"):
        d["note"] = "per-launch averages"
json.dump(summ, open(out + "/pmc_summary.json", "w"), indent=1)
print(json.dumps(summ, indent=1)[:3000])
PY
find $OUT -name "*.csv" -size +1M -delete
rm -rf $OUT/stats $OUT/pmc
head -5 $OUT/kernel_stats.csv
