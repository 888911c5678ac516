mkdir -p gpurun_out/r5j8
python -m pytest tests -x -q -m gpu > gpurun_out/r5j8/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5j8/pytest_gpu.log
tail -6 gpurun_out/r5j8/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
python bench.py --steps 20 --warmup 5 > gpurun_out/r5j8/bench.json 2> gpurun_out/r5j8/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r5j8/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5j8/bench.json"))
print({k: d[k] for k in ("value", "ms_per_step", "ms_per_step_first_allocated", "rel_err")})
print("roofline", d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
print("cpu", d["cpu_baseline"])
print("rel_err_detail", {k: v for k, v in d["rel_err_detail"].items() if k not in ("definition", "reference_noise_floor")})
s = d.get("secondary", {})
for k, v in s.items():
    if k == "converge":
        print(k, {b: (round(x["device_s"] * 1e3, 1), round(x["iteration_hbm_frac"], 3)) for b, x in v.items() if b != "workload"})
    elif k == "configs[2]":
        print(k, v["f32"]["ms_per_iteration"], v["f64"]["ms_per_iteration"])
    else:
        print(k, {a: b for a, b in v.items() if a != "workload" and a != "placement_tuning"})
print(d.get("secondary_error"), d.get("cpu_baseline_error"))
PY
