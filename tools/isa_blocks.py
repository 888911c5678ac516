#!/usr/bin/env python3
"""Instruction mix per basic block of a kernel in an assembly file written by tools/one_kernel.sh:
    python tools/isa_blocks.py /tmp/one_kernel_<tag>.s [kernel-name-substring] [min instructions]"""
import collections
import re
import sys


def blocks(path, sub="k_", minlen=40):
    lines = open(path).read().split("\n")
    out = []
    for i, l in enumerate(lines):
        if re.match(r"^_ZN5aoc\d\d.*:", l) and sub in l and "ILb" in l:
            end = next(j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm") or lines[j].startswith(".Lfunc_end"))
            cur = ["entry", []]
            bl = [cur]
            for l2 in lines[i + 1:end]:
                if re.match(r"^\.LBB\d+_\d+:", l2):
                    cur = [l2.split(":")[0], []]
                    bl.append(cur)
                elif l2.startswith("\t") and not l2.strip().startswith((";", ".")):
                    cur[1].append(l2.strip())
            out.append((l.split(":")[0], bl))
    return out


def kind(i):
    op = i.split()[0]
    if "f64" in op: return "f64"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "smem"
    if op.startswith("s_"): return "salu"
    if "accvgpr" in op: return "acc"
    return "valu32"


if __name__ == "__main__":
    sub = sys.argv[2] if len(sys.argv) > 2 else "k_"
    minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    for name, bl in blocks(sys.argv[1], sub):
        print(name)
        for lab, ins in bl:
            if len(ins) >= minlen:
                c = collections.Counter(kind(i) for i in ins)
                print("  %-10s %5d  %s" % (lab, len(ins), dict(sorted(c.items()))))
