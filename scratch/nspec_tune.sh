#!/bin/bash
for n in 1 2 3; do
  echo "== NSPEC=$n"
  AOC_NSPEC=$n python bench.py --steps 10 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['kernels_ms'].items()}, round(d['value']/1e6,2))"
done
