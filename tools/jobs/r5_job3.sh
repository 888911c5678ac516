set -o pipefail
python -m pytest tests -x -q -m "gpu and not slow" --deselect tests/test_gpu_sweep.py > gpurun_out/r5_t3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_t3.log
for w in 0 1; do
  echo "AOC_FW_WPE1=$w" >> gpurun_out/r5_small.log
  AOC_FW_WPE1=$w python tools/small_iter_time.py 4096 10 >> gpurun_out/r5_small.log 2>&1
  AOC_FW_WPE1=$w python tools/small_iter_time.py 1024 10 >> gpurun_out/r5_small.log 2>&1
  AOC_FW_WPE1=$w python tools/mpc_bench.py >> gpurun_out/r5_small.log 2>&1
done
timeout -k 10 400 python tools/vmm_lottery.py 131072 6 one,c64,c1024,torch rec > gpurun_out/r5_vmm_lottery.log 2>&1
tail -5 gpurun_out/r5_t3.log; grep -v amdgpu.ids gpurun_out/r5_small.log; tail -40 gpurun_out/r5_vmm_lottery.log
