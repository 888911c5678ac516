#!/bin/bash
# kernel trace of a converge-mode solve of 4096 trajectories: what a tail iteration (full Hessian, few tiles) consists of
OUT=gpurun_out/tail_trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/converge_timeline.py ${1:-4096} > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
python - $OUT/trace <<PY
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "aoc64::" in r["Kernel_Name"] or "k_s" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void ", "").replace("aoc64::", "").split("<")[0]
# iterations = from one backward kernel to the next; print iterations 5, 12, 25, 30 of the LAST solve in the trace
starts = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]).startswith(("k_backward", "k_bw_hcut")) and (i == 0 or not short(rows[i-1]["Kernel_Name"]).startswith(("k_bw_hcut", "k_track_hcut")))]
its = [rows[a:b] for a, b in zip(starts, starts[1:] + [len(rows)])]
last = its[-int('${2:-37}'):]
for n in (${3:-3, 12, 18, 25, 32}):
    if n >= len(last): continue
    it = last[n]
    t0 = int(it[0]["Start_Timestamp"])
    span = (int(it[-1]["End_Timestamp"]) - t0) / 1e3
    agg = collections.OrderedDict()
    for r in it:
        k = short(r["Kernel_Name"]); agg[k] = agg.get(k, 0) + (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    busy = sum(agg.values())
    print("iteration ~%d: span %.0f us, kernels %.0f us (gaps %.0f): %s" % (n, span, busy, span - busy, {k: round(v) for k, v in agg.items()}))
PY
