#!/bin/bash
# line-search policy sweep
for cfg in "1024 1" "1024 9" "1400 9" "2048 9" "1400 2" "768 9"; do
  set -- $cfg
  echo "== WCAP=$1 KGROW=$2"
  AOC_LS_WCAP=$1 AOC_LS_KGROW=$2 python bench.py --steps 10 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms'].items()})"
done
