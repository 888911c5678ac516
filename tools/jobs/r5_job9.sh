mkdir -p gpurun_out/r5j9
bash tools/mpc_trace.sh r5j9/mpc > gpurun_out/r5j9/mpc_trace.txt 2>&1; cat gpurun_out/r5j9/mpc_trace.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5j9/trace4096 -- python tools/small_iter_time.py 4096 20 > gpurun_out/r5j9/trace4096.log 2>&1
cp $(find gpurun_out/r5j9/trace4096 -name "*kernel_stats.csv" | head -1) gpurun_out/r5j9/small_4096x20_kernel_stats.csv
python -m pytest tests/test_gpu_edges.py -q -m gpu -k arena 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 --no-secondary > gpurun_out/r5j9/bench_box2.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/r5j9/bench_box2.json')); print('box2', d['ms_per_step'], d['ms_per_step_first_allocated'], d['placement_tuning']['two_stream_solver']['ms_per_iteration'], d['rel_err'], d['cpu_baseline']['value'])"
