# update (copy) pass of small batches by the number of horizon blocks per tile (build knob AOC_LSF_NY)
for G in 32 64 128 250; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_lsf$G.so
  [ $G = 32 ] && L=aircraftoptimalcontrol_amd/lib/libaoc_hip.so
  echo "== AOC_LSF_NY=$G"
  AOC_LIB=$L python tools/small_iter_time.py 4096 10 2>&1 | grep -v amdgpu
  AOC_LIB=$L python tools/small_iter_time.py 8192 10 2>&1 | grep -v amdgpu
  AOC_LIB=$L python tools/mpc_bench.py 2>&1 | grep -v amdgpu
done
