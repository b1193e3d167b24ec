#!/usr/bin/env python3
"""Copy the small artefacts of tools/profile_all.sh from gpurun_out/prof_<tag>/ into profiles/<tag>/ and rebuild
profiles/pmc_traffic.json (the per-launch HBM bytes bench.py reports as roofline.traffic) from their summaries.
usage: python tools/collect_profiles.py <tag>"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
dst = os.path.join(ROOT, "profiles", tag)
tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
for wl in sorted(os.listdir(src)):
    s = os.path.join(src, wl, "summary.json")
    if not os.path.exists(s):
        continue
    os.makedirs(os.path.join(dst, wl), exist_ok=True)
    for f in ("summary.json", "kernel_stats.csv"):
        if os.path.exists(os.path.join(src, wl, f)):
            shutil.copy(os.path.join(src, wl, f), os.path.join(dst, wl, f))
    d = json.load(open(s))
    if d.get("hbm_bytes_per_step"):
        traffic[wl] = {"hbm_bytes_per_launch": d["hbm_bytes_per_step"], "traffic_over_algorithmic": d.get("traffic_over_algorithmic"),
                       "source": f"profiles/{tag}/{wl}/summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, summed over the kernels of one step)"}
    k = {n: round(v["us_avg"], 1) for n, v in d.get("kernels", {}).items()}
    print(f"{wl:44s} step {d.get('step_us_rocprof', 0):8.1f} us  kernels {k}  traffic/alg {d.get('traffic_over_algorithmic')}")
json.dump(traffic, open(tpath, "w"), indent=1, sort_keys=True)
