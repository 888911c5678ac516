# pivoting of the boundary hops: swap code skipped when no lane of the wavefront swaps (thr1: same pivots as before), and
# threshold pivoting (thr8: the diagonal stays unless another entry of the column is more than 8 times larger)
for v in last13 thr1 thr8; do
  L=aircraftoptimalcontrol_amd/lib/variants/libaoc_$v.so
  echo "== $v"
  AOC_LIB=$L python -m pytest tests/test_gpu_hcut.py -q -m gpu -x 2>&1 | tail -2
  AOC_LIB=$L bash tools/mpc_trace.sh r5j20/$v | grep "hcut\|sum"
  AOC_LIB=$L python tools/small_iter_time.py 1024 10 2>&1 | grep -v amdgpu
  AOC_LIB=$L python tools/small_iter_time.py 4096 10 2>&1 | grep -v amdgpu
done
