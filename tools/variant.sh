#!/bin/bash
# build a kernel variant with extra -D flags ON THE GPU BOX and bench it (one stream, per-pass times):
#   variant.sh "<tag>" "<-D flags>" [env...]
TAG=$1; FLAGS=$2; shift 2
SO=/tmp/libaoc_$TAG.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $FLAGS aircraftoptimalcontrol_amd/csrc/aoc_kernels.hip -o $SO 2>/dev/null || { echo "build failed $TAG"; exit 0; }
echo -n "== $TAG [$FLAGS] $@ : "
env AOC_LIB=$SO "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), [(k['pass'][:4]+k['iterations'][:2], round(k['avg_ms'],3)) for k in d['kernels']])"
