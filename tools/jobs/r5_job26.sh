# the horizon cut beyond 64 tiles with fewer segments (tiles x segments <= 1024): Gauss-Newton iterations kk = 0..8
for B in 6144 8192 12288 16384 20480; do
  for S in 0 4 8 16; do
    nt=$(( (B + 63) / 64 ))
    [ $(( nt * S )) -le 1024 ] || continue
    echo "== B=$B (tiles $nt) AOC_BW_HCUT=$S"
    AOC_BW_HCUT=$S python tools/small_iter_time.py $B 9 2>&1 | grep -v amdgpu
  done
done
