#!/usr/bin/env python3
"""VGPR/AGPR liveness over the assembly of one kernel (tools/one_kernel.sh writes /tmp/one_kernel_<tag>.s): where in
the instruction stream the register pressure peaks, block by block.
    python tools/isa_liveness.py /tmp/one_kernel_<tag>.s [kernel-name-substring]
Prints per basic block: instructions, live registers on entry, maximum inside (and the index where it occurs)."""
import re
import sys

REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def parse(path, sub):
    lines = open(path).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if re.match(r"^_ZN5aoc\d\d.*:", l) and sub in l and "ILb" in l)
    i1 = next(j for j in range(i0, len(lines)) if lines[j].startswith(".Lfunc_end"))
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for l in lines[i0 + 1:i1]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = [l.split(":")[0], []]
            blocks.append(cur)
        elif l.startswith("\t") and not l.strip().startswith((";", ".")):
            cur[1].append(l.strip().split(";")[0].strip())
    return blocks


def defuse(ins):
    op, _, rest = ins.partition(" ")
    toks = [t.strip() for t in rest.split(",")] if rest else []
    if op.startswith("s_") and not op.startswith("s_waitcnt"):
        return set(), set()
    alluse = op.startswith(("global_store", "ds_write", "ds_store", "scratch_store", "buffer_store", "flat_store", "v_cmp", "v_cmpx",
                            "v_readlane", "v_readfirstlane", "global_atomic", "s_"))
    if alluse or not toks:
        u = set()
        for t in toks:
            u |= regs(t)
        return set(), u
    d = regs(toks[0])
    u = set()
    for t in toks[1:]:
        u |= regs(t)
    if op.startswith(("v_fmac", "v_mac", "v_swap", "v_pk_fmac")) or "_sdwa" in op or "dpp" in op:
        u |= d
    return d, u


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "k_"
    blocks = parse(path, sub)
    names = [b[0] for b in blocks]
    succ = {}
    for bi, (name, ins) in enumerate(blocks):
        s = set()
        fall = True
        for x in ins:
            op = x.split()[0]
            if op.startswith("s_cbranch"):
                s.add(x.split()[1])
            elif op == "s_branch":
                s.add(x.split()[1]); fall = False
            elif op == "s_endpgm":
                fall = False
        if fall and bi + 1 < len(blocks):
            s.add(names[bi + 1])
        succ[name] = s
    du = {name: [defuse(x) for x in ins] for name, ins in blocks}
    livein = {n: set() for n in names}
    changed = True
    while changed:
        changed = False
        for name, ins in reversed(blocks):
            live = set()
            for s in succ[name]:
                live |= livein.get(s, set())
            for d, u in reversed(du[name]):
                live = (live - d) | u
            if live != livein[name]:
                livein[name] = live; changed = True
    print("%-10s %6s %8s %8s %6s   (v/a at the peak)" % ("block", "instr", "live-in", "max", "at"))
    for name, ins in blocks:
        live = set()
        for s in succ[name]:
            live |= livein.get(s, set())
        best, at, bl = len(live), len(ins), live
        for idx in range(len(ins) - 1, -1, -1):
            d, u = du[name][idx]
            live = (live - d) | u
            if len(live) > best:
                best, at, bl = len(live), idx, set(live)
        if len(ins) >= 20:
            print("%-10s %6d %8d %8d %6d   v %d a %d   %s" % (name, len(ins), len(livein[name]), best, at,
                  sum(1 for r in bl if r[0] == "v"), sum(1 for r in bl if r[0] == "a"), ins[at] if at < len(ins) else ""))


if __name__ == "__main__":
    main()
