#!/bin/bash
REPO=$GRAFT_REPO_ROOT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/cls
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/cls -o p -- python3 $REPO/tools/convlstm_time.py > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, json
acc = {}
for path in glob.glob("/tmp/cls/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "convlstm_step_kernel" in r["Kernel_Name"]:
            key = (r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size"])
            acc.setdefault(key, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {}
for (k, g), m in acc.items():
    m = {c: sum(v) / len(v) for c, v in m.items()}
    wc = m["SQ_WAVE_CYCLES"]
    out[f"{k} grid={g}"] = {c[3:]: round(v / wc, 3) for c, v in m.items() if c != "SQ_WAVE_CYCLES"}
json.dump(out, open("/root/repo/gpurun_out/stall_convlstm.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
