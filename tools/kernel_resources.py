#!/usr/bin/env python3
"""Register / scratch / occupancy table of the library's kernels, from hipcc's kernel-resource-usage remarks.

    python tools/kernel_resources.py [-D flags ...] [--filter substr] > table

Compiles csrc/aoc_kernels.hip for gfx950 (no GPU needed) with -Rpass-analysis=kernel-resource-usage and prints one
line per kernel: VGPRs, AGPRs, SGPRs, scratch bytes per lane, waves per SIMD, LDS bytes, demangled name."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "aircraftoptimalcontrol_amd", "csrc", "aoc_kernels.hip")


def table(flags=(), keep_so=None):
    with tempfile.TemporaryDirectory() as d:
        so = keep_so or os.path.join(d, "lib.so")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-Rpass-analysis=kernel-resource-usage", *flags, SRC, "-o", so]
        err = subprocess.run(cmd, capture_output=True, text=True, check=True).stderr
    rows = []
    for b in re.split(r"remark: [^\n]*Function Name: ", err)[1:]:
        name = b.split("\n")[0].split()[0]
        g = lambda k: int(m.group(1)) if (m := re.search(k + r": (\d+)", b)) else -1
        rows.append([g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                     g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]"), name])
    dem = subprocess.run(["c++filt"], input="\n".join(r[-1] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for r, n in zip(rows, dem):
        r[-1] = re.sub(r"\(.*", "", n.replace("void ", ""))
    return rows


if __name__ == "__main__":
    args = sys.argv[1:]
    filt = None
    if "--filter" in args:
        i = args.index("--filter")
        filt = args[i + 1]
        del args[i:i + 2]
    print("%5s %5s %5s %7s %5s %6s  %s" % ("VGPR", "AGPR", "SGPR", "scratch", "waves", "LDS", "kernel"))
    for r in table(args):
        if filt is None or filt in r[-1]:
            print("%5d %5d %5d %7d %5d %6d  %s" % tuple(r))
