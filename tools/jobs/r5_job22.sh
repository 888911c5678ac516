# the segment kernels of the horizon cut on several wavefronts: bit-identity test, then timelines with the knob off / on
mkdir -p gpurun_out/r5j22
timeout -k 10 600 python -m pytest tests/test_gpu_hcut.py -q -m gpu -x > gpurun_out/r5j22/tests.log 2>&1; rc=$?; tail -5 gpurun_out/r5j22/tests.log
[ $rc = 0 ] || exit $rc
for w in 0 1; do
  echo "== AOC_HCUT_WAVES=$w"
  AOC_HCUT_WAVES=$w bash tools/mpc_trace.sh r5j22/waves$w | grep "hcut\|sum"
  AOC_HCUT_WAVES=$w python tools/small_iter_time.py 1024 10 2>&1 | grep -v amdgpu
  AOC_HCUT_WAVES=$w python tools/small_iter_time.py 1024 20 2>&1 | grep -v amdgpu
done
