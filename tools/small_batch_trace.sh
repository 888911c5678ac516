#!/bin/bash
# Where a small-batch iteration spends its time: bench line and kernel trace at 4096 trajectories, MPC step timing.
set -o pipefail
TAG=${1:-small}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python bench.py --batch-per-gpu 4096 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench4096.json 2> $OUT/bench4096.err || { tail -5 $OUT/bench4096.err; exit 1; }
timeout -k 10 200 python bench.py --batch-per-gpu 1024 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench1024.json 2> $OUT/bench1024.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --batch-per-gpu 4096 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python tools/trace_summary.py $OUT/trace 5 8 > $OUT/trace_summary.txt 2>&1
timeout -k 10 200 python tools/mpc_bench.py > $OUT/mpc.log 2>&1 || { tail -5 $OUT/mpc.log; exit 1; }
tail -3 $OUT/mpc.log
