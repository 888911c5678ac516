mkdir -p gpurun_out/r5j7
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_hcut.py -q -m gpu > gpurun_out/r5j7/hcut.log 2>&1; echo "hcut rc=$?" >> gpurun_out/r5j7/hcut.log
tail -15 gpurun_out/r5j7/hcut.log
for c in "0 0" "1 0" "1 2"; do set -- $c
  echo "AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2" >> gpurun_out/r5j7/small.log
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 4096 10 >> gpurun_out/r5j7/small.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 4096 20 >> gpurun_out/r5j7/small.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 1024 20 >> gpurun_out/r5j7/small.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/mpc_bench.py >> gpurun_out/r5j7/small.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r5j7/small.log
for d in perturbed random; do echo "== 4096 $d"; python tools/hcut_full_stats.py 4096 $d 14 2>&1 | grep -v amdgpu.ids; done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5j7/trace -- python tools/small_iter_time.py 4096 20 > gpurun_out/r5j7/trace.log 2>&1
f=$(find gpurun_out/r5j7/trace -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-90s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
