# bandwidth probe: the gains K~ stored in 4 / 6 bytes per entry instead of 8 (WRONG numerics for 4; timing only): per-pass times of the headline workload, one stream
for tag in base kt4 kt6; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_$tag.so
  [ $tag = base ] && L=aircraftoptimalcontrol_amd/lib/libaoc_hip.so
  echo "== $tag"
  for rep in 1 2; do
  AOC_LIB=$L timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap --placement-candidates 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), [(k['pass'][:4]+k['iterations'][:2], round(k['avg_ms'],3)) for k in d['kernels']])"
  done
done
