#!/bin/bash
# Run ON THE GPU BOX: does FETCH_SIZE count reads served by the Infinity Cache?  Config 3 (frame-sum pre-pass + simulation = every
# input byte read twice) at batch sizes whose input fits (16 clips = 134 MB) and does not fit (256 clips = 2.1 GB) the 256 MiB cache.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for B in 8 16 24 64 256; do
  rm -rf /tmp/mp_$B
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/mp_$B -o f -- python3 $REPO/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --workload cfg3_v2e_f32_256x32x256x256_bilinear5 --batch $B > /tmp/mp_$B.json 2>/dev/null
  python3 - $B /tmp/mp_$B /tmp/mp_$B.json << 'PY'
import csv, glob, json, sys
b, d, jf = int(sys.argv[1]), sys.argv[2], sys.argv[3]
acc = {}
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        k = "pre" if "shot_sum" in n else ("main" if "v2e_voxel_kernel" in n else None)
        if k and r["Counter_Name"] == "FETCH_SIZE":
            acc.setdefault(k, []).append(float(r["Counter_Value"]))
inp = b * 32 * 256 * 256 * 4
line = json.loads(open(jf).read().strip().splitlines()[-1])
print(f"B={b:4d} input {inp/1e6:8.1f} MB  kernel_ms {line['roofline']['kernel_ms_avg']:.4f}  " + "  ".join(
    f"{k}: FETCH_SIZE x2 = {2*1024*sum(v)/len(v)/1e6:8.1f} MB = {2*1024*sum(v)/len(v)/inp:.3f} x input" for k, v in sorted(acc.items())))
PY
done
