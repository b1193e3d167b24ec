#!/usr/bin/env python3
"""Per-kernel register / spill / LDS figures FROM THE CODE OBJECTS' OWN METADATA (the `.amdhsa` notes hipcc writes), not from a
profiler field: compiles every translation unit of libv2v_hip.so with `-S --cuda-device-only` (same flags as the Makefile) and
reads .vgpr_count / .vgpr_spill_count / .sgpr_count / .sgpr_spill_count / .group_segment_fixed_size per kernel.

    python tools/kernel_resources.py [out.json]        (default: profiles/kernel_resources.json; ~3 minutes, no GPU needed)
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "v2v_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"]
TUS = ["v2v_v2e_tu", "v2v_v2e_spec_f32_tu", "v2v_v2e_spec_u8_tu", "v2v_esim_u8_tu", "v2v_esim_f32_tu", "v2v_convlstm_tu", "v2v_capi"]
FIELDS = {".vgpr_count": "vgpr", ".vgpr_spill_count": "vgpr_spill", ".sgpr_count": "sgpr", ".sgpr_spill_count": "sgpr_spill",
          ".group_segment_fixed_size": "lds_static_bytes", ".private_segment_fixed_size": "scratch_bytes", ".agpr_count": "agpr"}


def parse(path):
    """One dict per kernel entry of `amdhsa.kernels` (an entry starts at the list item `  - .agpr_count: ...`; its fields come in
    alphabetical order, i.e. .agpr_count / .group_segment_fixed_size BEFORE .name -- attach them to the entry, not to the last name seen)."""
    out, cur = {}, None
    for line in open(path):
        item = re.match(r"  - (\.[a-z_]+):\s+(\S+)", line)              # kernel-level list item (argument items are indented deeper)
        m = item or re.match(r"    (\.[a-z_]+):\s+(\S+)", line)
        if not m:
            continue
        if item:
            cur = {}
        key, val = m.group(1), m.group(2)
        if cur is None:
            continue
        if key == ".name" and val.startswith("_Z"):
            out[val] = cur
        elif key in FIELDS:
            cur[FIELDS[key]] = int(val)
    return out


def main():
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "kernel_resources.json")
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for tu in TUS:
            s = os.path.join(tmp, tu + ".s")
            procs.append((tu, s, subprocess.Popen(["/opt/rocm/bin/hipcc", *FLAGS, "-o", s, os.path.join(CSRC, tu + ".hip")],
                                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
            if len(procs) % 4 == 0:
                for _, _, p in procs[-4:]:
                    p.wait()
        for tu, s, p in procs:
            p.wait()
            for name, r in parse(s).items():
                r["tu"] = tu
                res[name] = r
    names = list(res)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        res[n]["demangled"] = d
        alloc = (res[n].get("vgpr", 0) + res[n].get("agpr", 0) + 7) // 8 * 8
        res[n]["waves_per_simd_by_registers"] = min(8, 512 // alloc) if alloc else 8
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    spilled = {res[n]["demangled"]: res[n]["vgpr_spill"] for n in res if res[n].get("vgpr_spill")}
    print(f"{len(res)} kernels -> {dst}; kernels with VGPR spills: {len(spilled)}")
    for k, v in sorted(spilled.items(), key=lambda kv: -kv[1])[:12]:
        print(f"  {v:4d}  {k[:150]}")


main()
