# with the maps composed pairwise the chain is shorter: does another number of segments pay now?
# MPC step of 1024 instances and Gauss-Newton iterations of 1024 / 4096 trajectories by S (default 16 up to 64 tiles)
for S in 8 12 16 20 24 32; do
  echo -n "S=$S  "; AOC_BW_HCUT=$S AOC_TRACK_HCUT=$S python tools/mpc_bench.py 1024 500 60 2 2>&1 | grep -v amdgpu | cut -c1-110
done
for B in 1024 4096; do
  for S in 8 12 16 20 24 32; do
    echo -n "S=$S  "; AOC_BW_HCUT=$S python tools/small_iter_time.py $B 10 2>&1 | grep perturbed
  done
done
