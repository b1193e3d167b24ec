#!/usr/bin/env python3
"""Static scan of the built library for the instruction pattern behind round 5's co-scheduling finding (tools/pk_cohazard_probe.py): a
packed-float32 VALU instruction (v_pk_mul/add/fma_f32) that reads an SGPR pair, followed within a few instructions by a scalar
instruction that overwrites one of those SGPRs.  In the x2 upsampling kernel exactly such a site (`v_pk_mul_f32 .., s[0:1]` then
`s_mov_b32 s1, ..`) went wrong in lanes 48-63 whenever a matrix-core kernel of another stream shared the CU.

    python tools/pk_sgpr_war_scan.py [libv2v_hip.so] [window=12]
Prints every site; exit code 1 when any exists."""
import importlib.util
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def scan(so_path=None, window=12):
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(ROOT, "tools", "kernel_resources.py"))
    kr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kr)
    so_path = so_path or os.path.join(ROOT, "v2v_amd", "libv2v_hip.so")
    sites = []
    with tempfile.TemporaryDirectory() as tmp:
        for i, (_, blob) in enumerate(kr.carve_code_objects(so_path)):
            co = os.path.join(tmp, f"co{i}.elf")
            open(co, "wb").write(blob)
            dis = subprocess.run([os.path.join(kr.LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
            cur = None
            funcs = {}
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
                if m:
                    cur = m.group(1)
                    funcs[cur] = []
                elif cur and line.strip() and not line.strip().startswith(";"):
                    funcs[cur].append(line.split("//")[0].strip())
            for name, ins in funcs.items():
                for k, text in enumerate(ins):
                    if not re.match(r"v_pk_(mul|add|fma)_f32", text):
                        continue
                    regs = set()
                    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", text):
                        regs.update(range(int(a), int(b) + 1))
                    if not regs:
                        continue
                    for nxt in ins[k + 1:k + 1 + window]:
                        if re.match(r"s_(cbranch|branch|endpgm|setpc|swappc)", nxt):
                            break                                              # control flow: the straight-line window ends
                        if not nxt.startswith("s_") or re.match(r"s_(waitcnt|nop|barrier|cmp|bitcmp|sleep|setprio)", nxt):
                            continue
                        dst = nxt.split(None, 1)[1].split(",")[0].strip() if " " in nxt else ""
                        m1, m2 = re.match(r"s(\d+)$", dst), re.match(r"s\[(\d+):(\d+)\]$", dst)
                        written = {int(m1.group(1))} if m1 else set(range(int(m2.group(1)), int(m2.group(2)) + 1)) if m2 else set()
                        if written & regs:
                            sites.append((name, text, nxt))
                            break
    return sites


if __name__ == "__main__":
    found = scan(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else None, int(sys.argv[-1]) if sys.argv[-1].isdigit() else 12)
    names = subprocess.run(["c++filt"], input="\n".join(s[0] for s in found), capture_output=True, text=True).stdout.splitlines()
    for (n, a, b), d in zip(found, names):
        print(f"{d[:110]}\n    {a}\n    {b}")
    print(f"{len(found)} site(s)")
    sys.exit(1 if found else 0)
