#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel stats of any python script of this repo, top kernels by total time.  usage: profile_py.sh tools/x.py [args]
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/ppy
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ppy -o st -- python3 $REPO/"$@" > /tmp/ppy.log 2>&1
f=$(find /tmp/ppy -name "*kernel_stats.csv" | head -1)
python3 - "$f" << 'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%9.2f ms %6d x %8.1f us  min %7.1f max %7.1f  %s" % (float(r["TotalDurationNs"]) / 1e6, int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                              float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, r["Name"][:100]))
PY
