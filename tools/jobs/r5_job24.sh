# full GPU suite, MPC timeline, small batches, headline sanity on the build with the multi-wavefront segment kernels
mkdir -p gpurun_out/r5j24
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r5j24/pytest_gpu.log 2>&1; rc=$?; tail -3 gpurun_out/r5j24/pytest_gpu.log
[ $rc = 0 ] || exit $rc
bash tools/mpc_trace.sh r5j24/mpc_trace
python tools/mpc_bench.py 2>&1 | grep -v amdgpu
for B in 64 1024 2048 4096; do python tools/small_iter_time.py $B 20 2>&1 | grep -v amdgpu; done
python tools/converge_bench.py 4096 2>&1 | grep -v amdgpu | tail -3
