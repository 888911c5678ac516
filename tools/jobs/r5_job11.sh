# depth of the load rings of the one-wavefront passes (build knobs AOC_BW_PF / AOC_FW_PF) at 1024 and 2048 tiles
for tag in base bw2 bw3 fw3 fw4; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_$tag.so
  [ $tag = base ] && L=aircraftoptimalcontrol_amd/lib/libaoc_hip.so
  for B in 65536 131072; do
    echo "== $tag B=$B"
    AOC_LIB=$L timeout -k 10 120 python tools/vmm_lottery.py $B 3 torch rec 2>&1 | grep "^torch"
  done
done
