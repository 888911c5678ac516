#!/bin/bash
# Line-search schedules side by side on the bench workload (per-pass times of iterations 0..21), plus a kernel
# trace of the default schedule.  ls_ab.sh <tag>
set -o pipefail
TAG=${1:-ls_ab}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
AOC_LS_WORKLIST=0 timeout -k 10 120 python tools/late_iterations.py 131072 22 > $OUT/rounds.log 2>&1 || exit 1
timeout -k 10 120 python tools/late_iterations.py 131072 22 > $OUT/wl_cpl1.log 2>&1 || exit 1
AOC_LS_CPL=2 timeout -k 10 120 python tools/late_iterations.py 131072 22 > $OUT/wl_cpl2.log 2>&1 || exit 1
AOC_LS_WCAP=2048 timeout -k 10 120 python tools/late_iterations.py 131072 22 > $OUT/wl_cpl1_wcap2048.log 2>&1 || exit 1
AOC_LS_DEPTH_MIN=3 timeout -k 10 120 python tools/late_iterations.py 131072 22 > $OUT/wl_cpl1_depth3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python tools/late_iterations.py 131072 22 > $OUT/trace.log 2>&1 || exit 1
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
cp $(find $OUT/trace -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv
python tools/trace_summary.py $OUT/trace 1 5 10 17 20 > $OUT/trace_summary.txt 2>&1
for f in rounds wl_cpl1 wl_cpl2 wl_cpl1_wcap2048 wl_cpl1_depth3; do echo "== $f"; python - $OUT/$f.log <<'PY'
import sys,re
rows=[eval(re.search(r"ms (\[.*?\])", l).group(1)) for l in open(sys.argv[1]) if " ms [" in l]
ls=[r[2] for r in rows]
print("LS kk0-2 %.2f  kk3-8 %.2f  kk9-12 %.2f  kk13-15 %.2f  kk16-21 %.2f   mean iter(0..19) %.3f" % (sum(ls[0:3])/3, sum(ls[3:9])/6, sum(ls[9:13])/4, sum(ls[13:16])/3, sum(ls[16:22])/6, sum(sum(r) for r in rows[:20])/20))
PY
done
