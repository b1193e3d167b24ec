#!/usr/bin/env python3
"""Static instruction histogram of a kernel's hottest loop from `hipcc -S` output.

    python tools/isa_loop_hist.py file.s <substring of the mangled kernel name> [--dump]

Finds the function, takes the LARGEST backward-branch loop body (the unrolled time loop) and prints opcode counts by
class with the issue-cost weights measured in profiles/valu_rates_ubench.txt (cheap = 2 cycles, everything else 4)."""
import collections
import re
import sys

CHEAP = {"v_add_f32", "v_sub_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32",
         "v_mov_b32", "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_subrev_u32", "v_subrev_f32", "v_add_co_u32",
         "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_fmaak_f32", "v_fmamk_f32"}


def main():
    path, key = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            n = sum(1 for x in body[span[0]:span[1]] if re.match(r"\s+v_", x))
            if best is None or n > best[0]:
                best = (n, span)
    n, (a, b) = best
    ops = collections.Counter()
    for l in body[a:b + 1]:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if m and not l.strip().startswith((".", ";")):
            ops[m.group(1)] += 1
        if dump:
            print(l)
    valu = {k: v for k, v in ops.items() if k.startswith("v_")}
    cyc = sum(v * (2 if k.replace("_e32", "").replace("_e64", "") in CHEAP else 4) for k, v in valu.items())
    print(f"loop lines {a}..{b}: VALU {sum(valu.values())}  (weighted cycles {cyc})  SALU {sum(v for k, v in ops.items() if k.startswith('s_'))}"
          f"  DS {sum(v for k, v in ops.items() if k.startswith('ds_'))}  VMEM {sum(v for k, v in ops.items() if k.startswith(('global_', 'buffer_', 'scratch_')))}")
    for k, v in sorted(ops.items(), key=lambda kv: -kv[1]):
        print(f"  {k:28s} {v}")


main()
