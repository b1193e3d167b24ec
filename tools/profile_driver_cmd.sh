#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 --kernel-trace --stats of the DRIVER's exact bench command; keeps the kernel_stats.csv and the bench line.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_driver_cmd
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pdrv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pdrv -o st -- python3 $REPO/bench.py --gpus 1 --steps 20 --warmup 5 --extra-out $OUT/bench_extra_under_rocprof.json > $OUT/bench_line_under_rocprof.json 2> $OUT/err.log
cp $(find /tmp/pdrv -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
python3 - $OUT <<'PY'
import csv, json, sys
out = sys.argv[1]
d = json.loads(open(out + "/bench_line_under_rocprof.json").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
head = [r for r in rows if "esim_voxel_kernel<1, 4, 1, 1, true, false, false, true, false>" in r["Name"]]
print("bench line (HIP events): kernel_ms_avg %.4f  frac %.3f" % (d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
for r in head:
    print("rocprofv3 kernel stats  : %s calls, avg %.4f ms (min %.4f max %.4f)  %s" % (r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Name"][:90]))
PY
