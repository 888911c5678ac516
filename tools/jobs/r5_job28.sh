mkdir -p gpurun_out/r5j28
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r5j28/pytest_gpu.log 2>&1; rc=$?; tail -3 gpurun_out/r5j28/pytest_gpu.log
[ $rc = 0 ] || exit $rc
for B in 5120 8192; do python tools/small_iter_time.py $B 20 2>&1 | grep -v amdgpu; done
python tools/mpc_bench.py 8192 500 40 2 2>&1 | grep -v amdgpu
python tools/converge_bench.py 8192 2>&1 | grep -v amdgpu | tail -1 | cut -c1-400
