#!/usr/bin/env python3
"""Summarise the rocprofv3 CSVs of one bench.py workload (tools/profile_all.sh) into one small JSON: per-kernel average durations,
HBM traffic per STEP (FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950,
WRITE_SIZE as is; summed over the kernels one step launches), SQ instruction counts, and the bench line measured under the profiler."""
import csv
import glob
import json
import os
import sys

out_dir, workload = sys.argv[1], sys.argv[2]
KEYS = ("esim_voxel_kernel", "v2e_voxel_kernel", "v2e_shot_sum_kernel", "frontend_tile_kernel", "frontend_kernel", "count_pick_kernel",
        "normalize_pad_rows_kernel", "normalize_pad_kernel", "count_hist4_kernel", "clip_frames4_kernel")   # the hot path's kernels


def rows(pattern):
    for path in glob.glob(os.path.join(out_dir, pattern), recursive=True):
        with open(path, newline="") as f:
            for r in csv.DictReader(f):
                yield r


def short(name):
    for k in KEYS:
        if k in name:
            return k
    return None


# register / spill figures from the code objects' own metadata (tools/kernel_resources.py), keyed by the demangled kernel name
RES = {}
for cand in ("profiles/r05/kernel_resources.json", "profiles/r04/kernel_resources.json", "profiles/r03/kernel_resources.json", "profiles/kernel_resources.json"):
    rp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), cand)
    if os.path.exists(rp):
        RES = {v["demangled"]: v for v in json.load(open(rp)).values()}
        break


def code_object_resources(kernel_name):
    r = RES.get(kernel_name) or RES.get(kernel_name.replace(" [clone .kd]", ""))
    if r is None:
        return {}
    return {"instance": kernel_name, "vgpr_code_object": r.get("vgpr"), "vgpr_spill_code_object": r.get("vgpr_spill"), "sgpr_code_object": r.get("sgpr"),
            "waves_per_simd_by_registers": r.get("waves_per_simd_by_registers"), "lds_static_bytes_code_object": r.get("lds_static_bytes")}


summary = {"workload": workload}
bench = None
p = os.path.join(out_dir, "bench_under_stats.json")
if os.path.exists(p):
    try:
        d = json.loads([l for l in open(p) if l.startswith("{")][-1])
        bench = {"value": d["value"], "kernel_ms_avg": d["roofline"]["kernel_ms_avg"], "kernel_ms_p50": d["roofline"]["kernel_ms_p50"],
                 "frac": d["roofline"]["frac"], "algorithmic_bytes_per_launch": d["roofline"]["algorithmic_bytes_per_launch"],
                 "steps": d["steps"], "warmup": d["warmup"], "parity_check": d["parity_check"]}
        summary["bench"] = bench
    except Exception as e:  # noqa: BLE001
        summary["bench"] = f"unparsed: {e}"
# per-kernel durations from the trace; launches per step = dispatches / (steps + warmup) rounded
durs, meta = {}, {}
for r in rows("stats/**/*kernel_trace.csv"):
    k = short(r.get("Kernel_Name", ""))
    if k:
        durs.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        vg = r.get("VGPR_Count") or r.get("Arch_VGPR_Count")
        # the register figures come from the code object's metadata; rocprofv3's own field (in units of 2 on gfx950) is kept beside it
        meta[k] = {**code_object_resources(r.get("Kernel_Name", "")), "vgpr_field_of_rocprofv3": vg, "sgpr": r.get("SGPR_Count"), "lds_bytes": r.get("LDS_Block_Size"),
                   "grid": r.get("Grid_Size_X") or r.get("Grid_Size"), "workgroup": r.get("Workgroup_Size_X") or r.get("Workgroup_Size")}
n_steps = (bench["steps"] + bench["warmup"]) if isinstance(bench, dict) else None
summary["kernels"] = {}
step_us = 0.0
for k, v in durs.items():
    v.sort()
    per_step = round(len(v) / n_steps) if n_steps else None
    summary["kernels"][k] = {"dispatches": len(v), "per_step": per_step, "us_avg": sum(v) / len(v), "us_median": v[len(v) // 2], "us_min": v[0], "us_max": v[-1], **meta[k]}
    step_us += (sum(v) / len(v)) * (per_step or 1)
summary["step_us_rocprof"] = step_us


def counter_per_step(dirname, name):
    tot = {}
    for r in rows(f"{dirname}/**/*counter_collection.csv"):
        k = short(r.get("Kernel_Name", ""))
        if k and r.get("Counter_Name") == name:
            tot.setdefault(k, []).append(float(r["Counter_Value"]))
    out = 0.0
    for k, v in tot.items():
        per_step = summary["kernels"].get(k, {}).get("per_step") or 1
        out += (sum(v) / len(v)) * per_step
    return out if tot else None


fetch = counter_per_step("pmc_fetch", "FETCH_SIZE")
write = counter_per_step("pmc_write", "WRITE_SIZE")
if fetch is not None and write is not None:
    summary["FETCH_SIZE_raw_KiB_per_step"] = fetch
    summary["WRITE_SIZE_raw_KiB_per_step"] = write
    summary["hbm_bytes_per_step"] = fetch * 1024 * 2 + write * 1024      # FETCH_SIZE counts 64 B per 128-B request on gfx950: doubled
    if isinstance(bench, dict):
        summary["traffic_over_algorithmic"] = round(summary["hbm_bytes_per_step"] / bench["algorithmic_bytes_per_launch"], 4)
sq = {}
for name in ("SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
    v = counter_per_step("pmc_sq", name)
    if v is not None:
        sq[name] = v
summary["sq_counters_per_step"] = sq
print(json.dumps(summary, indent=1))
