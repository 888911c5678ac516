#!/bin/bash
# bench prebuilt kernel variants (variants/libaoc_<tag>.so, built in the container) at small batch sizes + the MPC loop:
#   variant_libs.sh OUTTAG "tag1 tag2 ..." B1 B2 ...
OUT=gpurun_out/$1; TAGS=$2; shift 2
mkdir -p $OUT
for TAG in $TAGS; do
  SO=$PWD/variants/libaoc_$TAG.so
  [ -f $SO ] || { echo "missing $SO"; continue; }
  for B in "$@"; do
    echo -n "== $TAG B=$B : " | tee -a $OUT/summary.txt
    AOC_LIB=$SO timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch-per-gpu $B 2>$OUT/err_${TAG}_$B.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k['pass'][:8]+k['iterations'][:2]:round(k['avg_ms'],3) for k in d['kernels']})" | tee -a $OUT/summary.txt
  done
  echo -n "== $TAG mpc : " | tee -a $OUT/summary.txt
  AOC_LIB=$SO timeout -k 10 200 python tools/mpc_bench.py 2>/dev/null | tail -1 | tee -a $OUT/summary.txt
done
