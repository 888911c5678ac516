mkdir -p gpurun_out/r5j17
python -m pytest tests -x -q -m gpu > gpurun_out/r5j17/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5j17/pytest_gpu.log
tail -5 gpurun_out/r5j17/pytest_gpu.log
bash tools/mpc_trace.sh r5j17/mpc > gpurun_out/r5j17/mpc_trace.txt 2>&1; cat gpurun_out/r5j17/mpc_trace.txt
for B in 1024 2048 4096; do python tools/small_iter_time.py $B 10 2>&1 | grep -v amdgpu; done
python tools/mpc_bench.py 2>&1 | grep -v amdgpu
