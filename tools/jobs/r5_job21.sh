# full GPU suite + timeline on the build with the stage inputs fetched ahead and the last segment at 1.3
mkdir -p gpurun_out/r5j21
python -m pytest tests -q -m gpu -x > gpurun_out/r5j21/pytest_gpu.log 2>&1; rc=$?; tail -3 gpurun_out/r5j21/pytest_gpu.log
[ $rc = 0 ] || exit $rc
bash tools/mpc_trace.sh r5j21/mpc_trace
python tools/mpc_bench.py 2>&1 | grep -v amdgpu
