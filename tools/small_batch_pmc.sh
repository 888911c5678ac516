#!/bin/bash
# SQ counters of the small-batch kernels (4096 trajectories): where the wavefront cycles go, instructions per wavefront,
# effective clock (GRBM_GUI_ACTIVE / 8 / duration).
set -o pipefail
TAG=${1:-small_pmc}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python bench.py --batch-per-gpu 4096 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python - $OUT <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/pmc/**/*counter_collection.csv", recursive=True)[0]
kt = glob.glob(d + "/pmc/**/*kernel_trace.csv", recursive=True)
res = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aoc64::", "")
    res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aoc64::", "")
        dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, dd in res.items():
    if not k.startswith(("k_backward", "k_forward", "k_ls_final", "k_track")):
        continue
    a = {c: sum(v) / len(v) for c, v in dd.items()}
    us = sum(dur[k]) / len(dur[k]) if dur.get(k) else float("nan")
    wc = a.get("SQ_WAVE_CYCLES", 0)
    print("%s  launches %d  avg %.1f us" % (k, len(next(iter(dd.values()))), us))
    print("   waves %d  VALU/wave %.0f (per stage %.1f)  LDS/wave %.0f (per stage %.1f)" % (
        a["SQ_WAVES"], a["SQ_INSTS_VALU"] / a["SQ_WAVES"], a["SQ_INSTS_VALU"] / a["SQ_WAVES"] / 499,
        a["SQ_INSTS_LDS"] / a["SQ_WAVES"], a["SQ_INSTS_LDS"] / a["SQ_WAVES"] / 499))
    print("   of wave cycles: parked (waitcnt/barrier) %.2f  issue-stalled %.2f  issuing %.2f (VALU %.2f)" % (
        a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc, a["SQ_ACTIVE_INST_VALU"] / wc))
    print("   wave cycles per wave (quad-cycles x4) %.0f  effective clock %.2f GHz" % (
        4 * wc / a["SQ_WAVES"], a.get("GRBM_GUI_ACTIVE", 0) / 8 / (us * 1e3) if us == us else float("nan")))
PY
