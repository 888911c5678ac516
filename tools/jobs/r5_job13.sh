mkdir -p gpurun_out/r5j13
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "seven_roles or linearisation_wavefront or split_kernels" > gpurun_out/r5j13/duo_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r5j13/duo_tests.log; tail -25 gpurun_out/r5j13/duo_tests.log
for d in 0 1 512; do
  echo "== AOC_FW_DUO=$d"
  AOC_FW_DUO=$d python tools/small_iter_time.py 1024 10 2>&1 | grep -v amdgpu
  AOC_FW_DUO=$d python tools/small_iter_time.py 2048 10 2>&1 | grep -v amdgpu
  AOC_FW_DUO=$d python tools/small_iter_time.py 4096 10 2>&1 | grep -v amdgpu
  AOC_FW_DUO=$d python tools/mpc_bench.py 2>&1 | grep -v amdgpu
done
