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
TRACE=$(find /tmp/pdrv -name "*kernel_trace.csv" | head -1)
python3 - $OUT $TRACE <<'PY'
import csv, json, sys
out = sys.argv[1]
# The headline instance is launched 5 (warm-up) + 20 (timed region) + 100 (settled tail, outside the timed region) times by the driver's
# command: kernel_stats.csv averages all 125.  What must agree with the line's HIP-event average is the TIMED REGION: dispatches 6..25 of
# that kernel in the kernel trace, in start order.
try:
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "esim_voxel_kernel<1, 4, 1, 1, true, false, false, true, false>" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    timed = dur[5:25]
    rec = {"kernel": "esim_voxel_kernel<1, 4, 1, 1, true, false, false, true, false>", "dispatches": len(dur),
           "timed_region_dispatches_6_to_25_avg_ms": sum(timed) / len(timed), "warmup_dispatches_1_to_5_avg_ms": sum(dur[:5]) / 5,
           "settled_tail_dispatches_26_on_avg_ms": sum(dur[25:]) / max(1, len(dur[25:])), "all_dispatches_avg_ms": sum(dur) / len(dur),
           "first_45_dispatches_ms": [round(v, 4) for v in dur[:45]],
           "gap_before_dispatch_us_first_45": [0.0] + [round((int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3, 1) for i in range(1, min(45, len(rows)))]}
    json.dump(rec, open(out + "/timed_region_from_trace.json", "w"), indent=1)
    print("rocprofv3 kernel trace  : timed region (dispatches 6..25) avg %.4f ms; warm-up %.4f; settled tail %.4f; all %d: %.4f" % (
        rec["timed_region_dispatches_6_to_25_avg_ms"], rec["warmup_dispatches_1_to_5_avg_ms"], rec["settled_tail_dispatches_26_on_avg_ms"], len(dur), rec["all_dispatches_avg_ms"]))
except Exception as exc:  # noqa: BLE001
    print("kernel trace not summarised:", exc)
d = json.loads(open(out + "/bench_line_under_rocprof.json").read().strip().splitlines()[-1])
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
head = [r for r in rows if "esim_voxel_kernel<1, 4, 1, 1, true, false, false, true, false>" in r["Name"]]
print("bench line (HIP events): kernel_ms_avg %.4f  frac %.3f" % (d["roofline"]["kernel_ms_avg"], d["roofline"]["frac"]))
for r in head:
    print("rocprofv3 kernel stats  : %s calls, avg %.4f ms (min %.4f max %.4f)  %s" % (r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Name"][:90]))
PY
