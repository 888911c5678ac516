mkdir -p gpurun_out/r5j6
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_hcut.py -q -m gpu > gpurun_out/r5j6/hcut.log 2>&1; echo "hcut rc=$?" >> gpurun_out/r5j6/hcut.log
tail -40 gpurun_out/r5j6/hcut.log
cat gpurun_out/hcut_full_level1.json gpurun_out/hcut_full_level2.json 2>/dev/null
for c in "0 0" "1 0" "1 1" "1 2"; do set -- $c
  echo "AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2" >> gpurun_out/r5j6/small.log
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 4096 10 >> gpurun_out/r5j6/small.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 4096 20 >> gpurun_out/r5j6/small.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 1024 20 >> gpurun_out/r5j6/small.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/mpc_bench.py >> gpurun_out/r5j6/small.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r5j6/small.log
AOC_BW_HCUT_FULL=2 python -m pytest tests/test_gpu_sweep.py -q -m "gpu and not slow" -k "test_parity_sweep and not large and not two_streams" > gpurun_out/r5j6/sweep_level2.log 2>&1; echo "sweep rc=$?" >> gpurun_out/r5j6/sweep_level2.log
tail -30 gpurun_out/r5j6/sweep_level2.log
