# the GPU suite with the multi-wavefront segment kernels forced everywhere the cut runs (also at 17-128 tiles)
mkdir -p gpurun_out/r5j29
AOC_HCUT_WAVES=2 timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r5j29/pytest_gpu_waves2.log 2>&1; rc=$?; tail -3 gpurun_out/r5j29/pytest_gpu_waves2.log
exit $rc
