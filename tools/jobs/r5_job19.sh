# length of the last segment of the horizon cut (build knob AOC_HC_LAST10, tenths of a regular segment) with the inputs fetched ahead
for f in 12 13 14; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_last$f.so
  echo "== AOC_HC_LAST10=$f"
  AOC_LIB=$L bash tools/mpc_trace.sh r5j19/last$f | grep "hcut\|sum"
  AOC_LIB=$L python tools/small_iter_time.py 1024 10 2>&1 | grep -v amdgpu
  AOC_LIB=$L python tools/small_iter_time.py 4096 10 2>&1 | grep -v amdgpu
done
