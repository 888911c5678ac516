#!/bin/bash
# Round evidence on the driver's command line (--steps 20 --warmup 5): the bench line, kernel-trace stats of the
# one-stream run (whose per-kernel averages must agree with the line's `kernels`) and of the default two-stream
# run, HBM traffic counters (separate FETCH_SIZE / WRITE_SIZE passes over the one-stream run).  The profiled runs take
# their buffers as first allocated (--placement-candidates 1): only the bench line itself chooses its allocation.
set -o pipefail
TAG=${1:-r04}
STEPS=${2:-20}
WARM=${3:-5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py --steps $STEPS --warmup $WARM > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
python bench.py --steps $STEPS --warmup $WARM --no-overlap --no-cpu-baseline --no-secondary --placement-candidates 1 > $OUT/bench_one_stream.json 2> $OUT/bench1.err || { tail -5 $OUT/bench1.err; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace1 -- python bench.py --steps $STEPS --warmup $WARM --no-overlap --no-cpu-baseline --no-secondary --placement-candidates 1 > $OUT/trace1.log 2>&1 || { tail -5 $OUT/trace1.log; exit 1; }
cp $(find $OUT/trace1 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_one_stream.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace2 -- python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-secondary --placement-candidates 1 > $OUT/trace2.log 2>&1 || { tail -5 $OUT/trace2.log; exit 1; }
cp $(find $OUT/trace2 -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_two_streams.csv
python tools/overlap_timeline.py $OUT/trace2 $OUT/overlap_two_streams.json > /dev/null || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python bench.py --steps $STEPS --warmup 0 --no-overlap --no-cpu-baseline --no-secondary --placement-candidates 1 > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python bench.py --steps $STEPS --warmup 0 --no-overlap --no-cpu-baseline --no-secondary --placement-candidates 1 > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
python tools/pmc_summary.py $OUT
echo "{\"steps\": $STEPS, \"warmup\": $WARM}" > $OUT/run.json
