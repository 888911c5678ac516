#!/bin/bash
# Evidence for the small configurations (BASELINE configs[1] and configs[4] at per-GPU size): bench lines at 1024 and
# 4096 trajectories, rocprofv3 kernel stats of the 4096 run, the receding-horizon loop and its kernel timeline.
set -o pipefail
TAG=${1:-small}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 1024 4096 8192; do
  timeout -k 10 300 python bench.py --batch-per-gpu $B --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$B.json 2> $OUT/bench_$B.err || { tail -5 $OUT/bench_$B.err; exit 1; }
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python bench.py --batch-per-gpu 4096 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_4096.csv
timeout -k 10 300 python tools/mpc_bench.py 1024 > $OUT/mpc_1024.log 2>&1 || { tail -5 $OUT/mpc_1024.log; exit 1; }
timeout -k 10 300 python tools/mpc_bench.py 8192 500 50 > $OUT/mpc_8192.log 2>&1 || { tail -5 $OUT/mpc_8192.log; exit 1; }
bash tools/mpc_trace.sh $TAG/mpc_trace > $OUT/mpc_trace.txt 2>&1
tail -1 $OUT/mpc_1024.log; tail -1 $OUT/mpc_8192.log; tail -12 $OUT/mpc_trace.txt
