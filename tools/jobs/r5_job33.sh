# maps of neighbouring segments composed pairwise before the chain (aoc_tuning.hcut_pairs): tests, then timelines off / on
mkdir -p gpurun_out/r5j33
timeout -k 10 600 python -m pytest tests/test_gpu_hcut.py tests/test_gpu_mpc.py -q -m gpu -x > gpurun_out/r5j33/tests.log 2>&1; rc=$?; tail -6 gpurun_out/r5j33/tests.log
[ $rc = 0 ] || exit $rc
for w in 0 1; do
  echo "== AOC_HCUT_PAIRS=$w"
  AOC_HCUT_PAIRS=$w bash tools/mpc_trace.sh r5j33/pairs$w | grep "hcut\|sum"
  AOC_HCUT_PAIRS=$w python tools/small_iter_time.py 1024 10 2>&1 | grep -v amdgpu
  AOC_HCUT_PAIRS=$w python tools/small_iter_time.py 4096 10 2>&1 | grep -v amdgpu
  AOC_HCUT_PAIRS=$w python tools/small_iter_time.py 8192 10 2>&1 | grep -v amdgpu
done
