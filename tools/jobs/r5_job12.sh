# crowded rounds of the work-list line search: constants pinned in VGPRs (216 registers, 2 wavefronts per SIMD) or not (135, 3)
for tag in base pinall pinnone; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_$tag.so
  [ $tag = base ] && L=aircraftoptimalcontrol_amd/lib/libaoc_hip.so
  echo "== $tag"
  AOC_LIB=$L python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --placement-candidates 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), [(k['pass'][:4]+k['iterations'][:2], round(k['avg_ms'],3)) for k in d['kernels']])"
  AOC_LIB=$L python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --placement-candidates 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two streams', round(d['ms_per_step'],3), d['placement_tuning']['two_stream_solver']['ms_per_iteration'])"
done
