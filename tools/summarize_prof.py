#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs written by tools/profile_gpu.sh into one small JSON (committed under profiles/)."""
import csv
import glob
import json
import os
import sys

out_dir, workload = sys.argv[1], sys.argv[2]
KERNEL = "esim_voxel_kernel"


def rows(pattern):
    for path in glob.glob(os.path.join(out_dir, pattern), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


summary = {"workload": workload, "kernel": KERNEL}
# ---- kernel stats
stats = [r for r in rows("stats/**/*kernel_stats.csv")]
summary["kernel_stats"] = [{k: r[k] for k in r} for r in stats if KERNEL in r.get("Name", "") or "synth" in r.get("Name", "")]
# ---- per-dispatch durations from the kernel trace
durs = []
for r in rows("stats/**/*kernel_trace.csv"):
    if KERNEL in r.get("Kernel_Name", ""):
        durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        summary["vgpr"] = r.get("VGPR_Count") or r.get("Arch_VGPR_Count")
        summary["sgpr"] = r.get("SGPR_Count")
        summary["lds_bytes"] = r.get("LDS_Block_Size")
        summary["grid"] = r.get("Grid_Size_X") or r.get("Grid_Size")
        summary["workgroup"] = r.get("Workgroup_Size_X") or r.get("Workgroup_Size")
if durs:
    durs.sort()
    tail = durs[3:] if len(durs) > 6 else durs      # drop warm-up launches
    summary["dispatches"] = len(durs)
    summary["kernel_us_avg"] = sum(durs) / len(durs)
    summary["kernel_us_median"] = durs[len(durs) // 2]
    summary["kernel_us_min"] = durs[0]
    summary["kernel_us_max"] = durs[-1]


def counter(dirname, name):
    vals = []
    for r in rows(f"{dirname}/**/*counter_collection.csv"):
        if KERNEL in r.get("Kernel_Name", "") and r.get("Counter_Name") == name:
            vals.append(float(r["Counter_Value"]))
    return vals


fetch = counter("pmc_fetch", "FETCH_SIZE")
write = counter("pmc_write", "WRITE_SIZE")
if fetch:
    f = sum(fetch) / len(fetch)
    summary["FETCH_SIZE_raw_KiB_per_launch"] = f
    # gfx950: FETCH_SIZE reports exactly half of a wide (16 B/lane) coalesced streaming read (MI355X_MICROARCH.md, HBM)
    summary["fetch_bytes_per_launch_corrected"] = f * 1024 * 2
if write:
    w = sum(write) / len(write)
    summary["WRITE_SIZE_raw_KiB_per_launch"] = w
    summary["write_bytes_per_launch"] = w * 1024
if fetch and write:
    summary["hbm_bytes_per_launch"] = summary["fetch_bytes_per_launch_corrected"] + summary["write_bytes_per_launch"]
sq = {}
for name in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY",
             "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT"):
    v = counter("pmc_sq", name)
    if v:
        sq[name] = sum(v) / len(v)
for name in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_ACTIVE_INST_LDS", "SQ_LDS_IDX_ACTIVE",
             "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
    v = counter("pmc_sq2", name)
    if v:
        sq[name] = sum(v) / len(v)
for name in ("GRBM_GUI_ACTIVE", "GRBM_COUNT"):
    v = counter("pmc_grbm", name)
    if v:
        sq[name] = sum(v) / len(v)
summary["sq_counters_per_launch"] = sq
for tag in ("stats", "pmc_fetch", "pmc_write", "pmc_sq"):
    p = os.path.join(out_dir, f"bench_under_{tag}.json")
    if os.path.exists(p):
        try:
            line = [l for l in open(p) if l.startswith("{")][-1]
            d = json.loads(line)
            summary[f"bench_under_{tag}"] = {"value": d["value"], "kernel_ms_avg": d["roofline"]["kernel_ms_avg"],
                                             "achieved_GBps": d["roofline"]["achieved"], "frac": d["roofline"]["frac"],
                                             "algorithmic_bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"]}
        except Exception as e:
            summary[f"bench_under_{tag}"] = f"unparsed: {e}"
print(json.dumps(summary, indent=1))
