# 65-128 tiles with 8 segments: full-Hessian iterations and the receding-horizon step
for B in 6144 8192; do
  for S in 0 8; do
    echo "== B=$B AOC_BW_HCUT=$S AOC_TRACK_HCUT=$S"
    AOC_BW_HCUT=$S python tools/small_iter_time.py $B 20 2>&1 | grep -v amdgpu
    AOC_BW_HCUT=$S AOC_TRACK_HCUT=$S python tools/mpc_bench.py $B 500 40 2 2>&1 | grep -v amdgpu
  done
done
for B in 5120 7168; do
  for S in 0 8; do
    echo "== B=$B AOC_BW_HCUT=$S"
    AOC_BW_HCUT=$S python tools/small_iter_time.py $B 9 2>&1 | grep -v amdgpu
  done
done
