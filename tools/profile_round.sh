#!/bin/bash
# Round evidence: bench line, kernel-trace stats, HBM traffic counters (separate passes).
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps 3 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
python tools/pmc_summary.py $OUT
