# where the pairwise composition stops paying: ms per Gauss-Newton iteration by batch size, AOC_HCUT_PAIRS = 0 / 1
for B in 1536 2048 3072 4096; do
  for w in 0 1; do
    echo -n "pairs=$w  "; AOC_HCUT_PAIRS=$w python tools/small_iter_time.py $B 10 2>&1 | grep perturbed
  done
done
for B in 2048 4096; do for w in 0 1; do echo -n "pairs=$w  "; AOC_HCUT_PAIRS=$w AOC_TRACK_HCUT=16 python tools/mpc_bench.py $B 500 40 2 2>&1 | grep -v amdgpu | cut -c1-110; done; done
