# end of round 5: GPU suite, smoke, the driver's bench command, MPC timeline — on HEAD
mkdir -p gpurun_out/r5final
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r5final/pytest_gpu.log 2>&1; rc=$?; tail -3 gpurun_out/r5final/pytest_gpu.log
[ $rc = 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3
python bench.py --steps 20 --warmup 5 > gpurun_out/r5final/bench.json 2> gpurun_out/r5final/bench.err; rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5final/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "rel_err")}, d["roofline"]["frac"], d["cpu_baseline"]["value"])
for k, v in d.get("secondary", {}).items():
    if isinstance(v, dict): print(k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if isinstance(b, (int, float))})
PY
[ $rc = 0 ] || exit $rc
bash tools/mpc_trace.sh r5final/mpc_trace
