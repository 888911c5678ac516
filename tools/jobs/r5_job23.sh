# last-segment length (AOC_HC_LAST10) with the multi-wavefront segment kernels (16 tiles) and the one-wavefront ones (64 tiles)
for f in 10 11 12; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_mw$f.so
  echo "== AOC_HC_LAST10=$f"
  AOC_LIB=$L bash tools/mpc_trace.sh r5j23/mw$f | grep "hcut\|sum"
  AOC_LIB=$L python tools/small_iter_time.py 1024 10 2>&1 | grep -v amdgpu
  AOC_LIB=$L python tools/small_iter_time.py 4096 10 2>&1 | grep -v amdgpu
done
echo "== 13 (library)"
python tools/small_iter_time.py 4096 10 2>&1 | grep -v amdgpu
