#!/bin/bash
# rocprofv3 kernel stats for the secondary kernels (v2e, front-end pipeline, noise-on, uint8, post-ops); run on the GPU box.
set -u
TAG=${1:-r01c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_others_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for wl in cfg2_noise_on cfg2_noise_on_fast cfg2_u8 cfg3_v2e_f32_256x32x256x256_bilinear5 cfg4_pipeline_720p_to_256_41f_sum5 cfg4_u8_256x41x256x256_sum5 train_u8_12x201x128x128_sum5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$wl -o s -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --workload $wl > $OUT/$wl.json 2> $OUT/$wl.err
done
cat > $OUT/postops_run.py <<PY
import torch, sys
sys.path.insert(0, "$REPO")
from v2v_amd import postops, voxel
v = torch.round(torch.randn((12, 40, 5, 128, 128), device="cuda") * 3)
for _ in range(5):
    postops.normalize_and_pad(v, True, 16)
import numpy as np
g = np.random.default_rng(0); n = 2_000_000
ts = np.sort(g.uniform(0, 0.05, n)); xs = g.integers(0, 240, n); ys = g.integers(0, 180, n); ps = g.integers(0, 2, n)
ev = [torch.from_numpy(a).cuda() for a in (ts, xs, ys, ps)]
for _ in range(5):
    voxel.make_voxel(ev, 180, 240, 5, True)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/postops_events -o s -- python3 $OUT/postops_run.py > /dev/null 2> $OUT/postops_events.err
cd $REPO
python3 - <<PY
import csv, glob, json, os
out = {}
for d in sorted(glob.glob("$OUT/*/")):
    name = os.path.basename(d.rstrip("/"))
    rows = []
    for p in glob.glob(d + "**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if "v2v::" in r["Name"]:
                rows.append({"kernel": r["Name"][:90], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3})
    out[name] = rows
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
find $OUT -name "*.csv" -size +1M -delete
