mkdir -p gpurun_out/r5j14
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for tag in base b3; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_$tag.so; D=1
  [ $tag = base ] && L=aircraftoptimalcontrol_amd/lib/libaoc_hip.so
  [ $tag = split ] && L=aircraftoptimalcontrol_amd/lib/libaoc_hip.so && D=0
  AOC_LIB=$L AOC_FW_DUO=$D timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5j14/$tag -- python tools/mpc_bench.py 1024 500 60 2 > gpurun_out/r5j14/$tag.log 2>&1
  f=$(find gpurun_out/r5j14/$tag -name "*kernel_stats.csv" | head -1)
  echo "== $tag: $(grep 'ms per' gpurun_out/r5j14/$tag.log | sed 's/.*= \([0-9.]* ms per receding-horizon step\).*/\1/')"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k_forward" in r["Name"] or "k_mpc_warm" in r["Name"]:
        print("   %-60s calls %5s avg %8.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
