#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of the two channels-last config-5 workloads (stock / fused consumer), top kernels by time.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
mkdir -p $REPO/gpurun_out/prof_cfg5
for wl in ${1:-cfg5_channels_last cfg5_fused_convlstm_channels_last}; do
  rm -rf /tmp/p5_$wl
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5_$wl -o st -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --hip-graph off --workload $wl > $REPO/gpurun_out/prof_cfg5/$wl.json 2>/dev/null
  f=$(find /tmp/p5_$wl -name "*kernel_stats.csv" | head -1)
  cp "$f" $REPO/gpurun_out/prof_cfg5/${wl}_kernel_stats.csv
  echo "== $wl"; tail -1 $REPO/gpurun_out/prof_cfg5/$wl.json | cut -c1-140
  python3 - "$f" << 'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {int(r["Calls"]):6d} x {float(r["AverageNs"])/1e3:8.1f} us  {r["Name"][:150]}')
PY
done
