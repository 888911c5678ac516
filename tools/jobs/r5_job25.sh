mkdir -p gpurun_out/r5j25
python bench.py --steps 20 --warmup 5 > gpurun_out/r5j25/bench.json 2> gpurun_out/r5j25/bench.err; rc=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5j25/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "rel_err")})
print("roofline", d["roofline"])
print("cpu", d["cpu_baseline"])
for k, v in d.get("secondary", {}).items():
    print(k, {a: b for a, b in v.items() if not isinstance(b, (dict, list))} if isinstance(v, dict) else v)
PY
exit $rc
