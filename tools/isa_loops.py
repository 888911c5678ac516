#!/usr/bin/env python3
"""Loops of a kernel in an assembly file written by tools/one_kernel.sh: for every innermost loop the instructions of
its body by kind (the stage loops of the roles of a multi-wavefront kernel are separate loops).
    python tools/isa_loops.py /tmp/one_kernel_<tag>.s [kernel-name-substring]"""
import collections
import re
import sys

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from isa_blocks import kind


def loops(path, sub):
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if re.match(r"^_ZN5aoc\d\d.*:", l) and sub in l and "ILb" in l)
    i1 = next(j for j in range(i0, len(lines)) if lines[j].startswith(".Lfunc_end"))
    cur_hdr, out, order = None, collections.OrderedDict(), []
    label = "entry"
    for l in lines[i0 + 1:i1]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*;(.*)", l)
        if m:
            label = m.group(1)
            c = m.group(2)
            h = re.search(r"Header=BB(\d+_\d+)", c)
            if "Loop Header" in c and "Inner" in c or ("=>This" in c and "Inner Loop Header" in c):
                cur_hdr = label
            elif h:
                cur_hdr = ".LBB" + h.group(1)
            else:
                cur_hdr = None
            if cur_hdr and cur_hdr not in out:
                out[cur_hdr] = []
            continue
        if re.match(r"^\.LBB\d+_\d+:", l):
            label = l.split(":")[0]
            cur_hdr = None
            continue
        if cur_hdr and l.startswith("\t") and not l.strip().startswith((";", ".")):
            out[cur_hdr].append(l.strip())
    return out


if __name__ == "__main__":
    for hdr, ins in loops(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "k_").items():
        c = collections.Counter(kind(i) for i in ins)
        print("loop %-10s %5d instructions  %s" % (hdr, len(ins), dict(sorted(c.items()))))
