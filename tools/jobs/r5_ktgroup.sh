# K~ slabs of G tiles interleaved (build knob AOC_KT_GROUP): per-pass times of six solvers built in a row (torch allocator)
for G in 1 2 8 32; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_ktg$G.so
  [ $G = 1 ] && L=aircraftoptimalcontrol_amd/lib/libaoc_hip.so
  echo "== AOC_KT_GROUP=$G"
  AOC_LIB=$L timeout -k 10 200 python tools/vmm_lottery.py 131072 6 torch rec 2>&1 | grep "^torch"
done
