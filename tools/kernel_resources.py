#!/usr/bin/env python3
"""Per-kernel register / spill / LDS figures FROM THE CODE OBJECTS' OWN METADATA (the `.amdhsa` notes hipcc writes), not from a
profiler field: compiles every translation unit of libv2v_hip.so with `-S --cuda-device-only` (same flags as the Makefile) and
reads .vgpr_count / .vgpr_spill_count / .sgpr_count / .sgpr_spill_count / .group_segment_fixed_size per kernel.

    python tools/kernel_resources.py [out.json]        (default: profiles/kernel_resources.json; ~3 minutes, no GPU needed)
    python tools/kernel_resources.py --from-so [out.json]   the same fields out of the BUILT v2v_amd/libv2v_hip.so (carves the gfx950 code
                                                       objects out of .hip_fatbin, reads their notes and disassembly; seconds)

`from_so()` is what tests/test_kernel_resources.py runs: no kernel of the shipped library may use scratch memory or spill registers
outside an explicit allow-list (round 4 shipped 560 bytes of scratch in one ESIM instance and nothing looked).
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "v2v_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-S", "--cuda-device-only"]
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


LLVM = "/opt/rocm/lib/llvm/bin"


def carve_code_objects(so_path):
    """The device ELFs inside a host library's .hip_fatbin: a sequence of uncompressed clang offload bundles ('__CLANG_OFFLOAD_BUNDLE__',
    u64 entry count, per entry {u64 offset, u64 size, u64 triple length, triple}); -> [(triple, bytes)] of the amdgcn entries."""
    import struct
    data = open(so_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], data.find(magic)
    while pos >= 0:
        n, = struct.unpack_from("<Q", data, pos + len(magic))
        q = pos + len(magic) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "amdgcn" in triple and size:
                out.append((triple, data[pos + off:pos + off + size]))
        pos = data.find(magic, pos + len(magic))
    return out


def from_so(so_path=None):
    """{mangled kernel name: {vgpr, vgpr_spill, sgpr, sgpr_spill, lds_static_bytes, scratch_bytes, agpr, scratch_instructions, arch}} of the
    built library, from the code objects' own notes (llvm-readelf --notes) and their disassembly (count of scratch_* instructions)."""
    so_path = so_path or os.path.join(ROOT, "v2v_amd", "libv2v_hip.so")
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for i, (triple, blob) in enumerate(carve_code_objects(so_path)):
            co = os.path.join(tmp, f"co{i}.elf")
            open(co, "wb").write(blob)
            notes = os.path.join(tmp, f"co{i}.notes")
            open(notes, "w").write(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout)
            kernels = parse(notes)
            # scratch_* instructions per function symbol of the disassembly
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
            cur, scratch = None, {}
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                elif cur and re.search(r"\bscratch_(load|store)", line):
                    scratch[cur] = scratch.get(cur, 0) + 1
            pk, curp = {}, None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    curp = m.group(1)
                elif curp and re.search(r"\bv_pk_(mul|add|fma)_f32", line):
                    pk[curp] = pk.get(curp, 0) + 1
            for name, r in kernels.items():
                r["packed_f32_instructions"] = pk.get(name, 0)
                r["arch"] = triple.split("--")[-1] if "--" in triple else triple
                r["scratch_instructions"] = scratch.get(name, 0)
                r["code_object"] = i
                res[name] = r
    names = list(res)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    for n, d in zip(names, dem):
        res[n]["demangled"] = d
    return res


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--from-so":
        res = from_so()
        dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "kernel_resources.json")
        json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
        bad = {r["demangled"]: (r.get("scratch_bytes", 0), r.get("vgpr_spill", 0), r["scratch_instructions"]) for r in res.values()
               if r.get("scratch_bytes", 0) or r.get("vgpr_spill", 0) or r["scratch_instructions"]}
        print(f"{len(res)} kernels of the built library -> {dst}; with scratch / VGPR spills: {len(bad)}")
        for k, v in bad.items():
            print(f"  scratch {v[0]} B, {v[1]} spilled VGPRs, {v[2]} scratch instructions  {k[:140]}")
        return
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


if __name__ == "__main__":
    main()
