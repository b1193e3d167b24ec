#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per kernel: python tools/pmc_kernels.py <dir> -> one line per (kernel, counter)."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = (row["Kernel_Name"].split("(")[0][:70], row["Counter_Name"])
            acc[k][0] += float(row["Counter_Value"])
            acc[k][1] += 1
for (kern, ctr), (tot, n) in sorted(acc.items()):
    print(f"{kern:72s} {ctr:24s} launches={n:4d} avg={tot / n:.4g}")
