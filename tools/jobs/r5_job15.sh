for d in 0 1; do
  echo "== AOC_FW_DUO=$d"
  for B in 64 512 1024 2048 2688; do AOC_FW_DUO=$d python tools/small_iter_time.py $B 10 2>&1 | grep -v amdgpu; done
  AOC_FW_DUO=$d python tools/small_iter_time.py 1024 20 2>&1 | grep -v amdgpu
  AOC_FW_DUO=$d python tools/mpc_bench.py 2>&1 | grep -v amdgpu
done
python -m pytest tests/test_gpu_parity.py tests/test_gpu_mpc.py tests/test_gpu_hcut.py tests/test_gpu_edges.py tests/test_gpu_pertraj.py -q -m gpu 2>&1 | tail -3
