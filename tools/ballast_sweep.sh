#!/bin/bash
# tools/ballast_probe.py for a list of ballast sizes, one fresh process each: where in the device memory (as far as the order
# of allocation decides it) does a solver run fast?   tools/ballast_sweep.sh "0 64 128 ..." [repeats]
OUT=gpurun_out/ballast
mkdir -p $OUT
for rep in $(seq 1 ${2:-1}); do
  for g in $1; do
    python tools/ballast_probe.py $g 2>/dev/null | grep ballast | tee -a $OUT/sweep.txt || exit 1
  done
done
