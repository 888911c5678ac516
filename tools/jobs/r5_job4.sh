mkdir -p gpurun_out
python -m pytest tests/test_gpu_hcut.py -x -q -m gpu > gpurun_out/r5_t4_hcut.log 2>&1; echo "hcut rc=$?" >> gpurun_out/r5_t4_hcut.log
tail -25 gpurun_out/r5_t4_hcut.log
for c in "0 0" "1 0" "1 1"; do set -- $c
  echo "AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2" >> gpurun_out/r5_small2.log
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 4096 10 >> gpurun_out/r5_small2.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 4096 20 >> gpurun_out/r5_small2.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/small_iter_time.py 1024 20 >> gpurun_out/r5_small2.log 2>&1
  AOC_HCUT_CHAIN6=$1 AOC_BW_HCUT_FULL=$2 python tools/mpc_bench.py >> gpurun_out/r5_small2.log 2>&1
done
grep -v amdgpu.ids gpurun_out/r5_small2.log
bash tools/jobs/r5_ktgroup.sh > gpurun_out/r5_ktgroup.log 2>&1; cat gpurun_out/r5_ktgroup.log
