#!/bin/bash
# build kernel variants with extra -D flags ON THE GPU BOX and bench them at small batch sizes:
#   variant_small.sh "<tag>" "<-D flags>" B1 B2 ...
TAG=$1; FLAGS=$2; shift 2
SO=/tmp/libaoc_$TAG.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $FLAGS aircraftoptimalcontrol_amd/csrc/aoc_kernels.hip -o $SO 2>/dev/null || { echo "build failed $TAG"; exit 0; }
for B in "$@"; do
echo -n "== $TAG [$FLAGS] B=$B : "
AOC_LIB=$SO python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch-per-gpu $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), {k['pass'][:8]+k['iterations'][:2]:round(k['avg_ms'],3) for k in d['kernels']})"
done
