# stage inputs fetched one stage ahead in the horizon-cut kernels: tests, then the per-kernel timeline of an MPC step and the small batches
mkdir -p gpurun_out/r5j18
python -m pytest tests/test_gpu_hcut.py tests/test_gpu_mpc.py -q -m gpu -x > gpurun_out/r5j18/tests.log 2>&1; rc=$?; tail -5 gpurun_out/r5j18/tests.log
[ $rc = 0 ] || exit $rc
bash tools/mpc_trace.sh r5j18/mpc_trace && cat gpurun_out/r5j18/mpc_trace/*.txt 2>/dev/null
python tools/mpc_bench.py 2>&1 | grep -v amdgpu
python tools/small_iter_time.py 1024 10 2>&1 | grep -v amdgpu
python tools/small_iter_time.py 4096 10 2>&1 | grep -v amdgpu
