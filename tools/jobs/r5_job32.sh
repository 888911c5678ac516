# HBM traffic of the small-batch kernels at 4096 trajectories (configs[1]): FETCH_SIZE / WRITE_SIZE per launch, and the kernel times
OUT=gpurun_out/r5j32
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
CMD="python bench.py --batch-per-gpu 4096 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/f.log 2>&1 || { tail -5 $OUT/f.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/w.log 2>&1 || { tail -5 $OUT/w.log; exit 1; }
python tools/pmc_summary.py $OUT | head -12
python - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"].split("(")[0].replace("void aoc64::", "")[:60].ljust(60), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), r["Percentage"])
PY
