#!/bin/bash
# kernel timeline of receding-horizon steps (1024 instances): which kernels a step consists of and the gaps between them
set -o pipefail
OUT=gpurun_out/${1:-mpc_trace}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python tools/mpc_bench.py 1024 500 40 2 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/trace/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(r['Kernel_Name'].split('(')[0].replace('void ', '').replace('aoc64::', ''), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
# a step starts with the tracking gains (k_track_gains*, or k_track_hcut_map with the horizon cut)
idx = [i for i, s in enumerate(seq) if s[0].startswith(('k_track_gains', 'k_track_hcut_map'))]
steps = [seq[a:b] for a, b in zip(idx[:-1], idx[1:])]
steps = steps[len(steps) // 2:]   # warm part
agg = collections.OrderedDict(); gaps = 0.0; span = 0.0
for st in steps:
    span += (st[-1][2] - st[0][1]) / 1e3
    for j, s in enumerate(st):
        agg[s[0]] = agg.get(s[0], 0) + (s[2] - s[1]) / 1e3
        if j: gaps += max(0, (s[1] - st[j - 1][2]) / 1e3)
n = len(steps)
print("steps averaged: %d, kernels per step: %d" % (n, len(steps[0])))
for k, v in agg.items(): print("  %-50s %8.1f us" % (k[:50], v / n))
print("  sum of kernels %.1f us, gaps %.1f us, span %.1f us" % (sum(agg.values()) / n, gaps / n, span / n))
PY
