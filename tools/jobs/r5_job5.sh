mkdir -p gpurun_out/r5j5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in perturbed random; do for B in 4096 1024; do echo "== $B $d"; python tools/hcut_full_stats.py $B $d 14 2>&1 | grep -v amdgpu.ids; done; done > gpurun_out/r5j5/stats.txt
cat gpurun_out/r5j5/stats.txt
for c in 0 1; do
  AOC_HCUT_CHAIN6=$c timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5j5/trace_c$c -- python tools/small_iter_time.py 4096 20 > gpurun_out/r5j5/trace_c$c.log 2>&1
  f=$(find gpurun_out/r5j5/trace_c$c -name "*kernel_stats.csv" | head -1)
  echo "== chain6=$c  $f"; python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%-90s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
