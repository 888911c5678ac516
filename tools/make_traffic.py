"""profiles/<name>_* from a gpurun_out/<tag> directory produced by tools/profile_round.sh, and profiles/traffic.json
(what bench.py reports as roofline.traffic / kernels[].traffic when batch, horizon and --steps are those of the run)."""
import json, shutil, sys, os
tag = sys.argv[1]; name = sys.argv[2] if len(sys.argv) > 2 else tag
src = "gpurun_out/" + tag
os.makedirs("profiles", exist_ok=True)
for f, g in (("kernel_stats_one_stream.csv", "kernel_stats_one_stream.csv"), ("kernel_stats_two_streams.csv", "kernel_stats_two_streams.csv"),
             ("pmc_summary.json", "pmc_summary.json"), ("bench.json", "bench.json"), ("bench_one_stream.json", "bench_one_stream.json")):
    shutil.copy(os.path.join(src, f), "profiles/%s_%s" % (name, g))
p = json.load(open(src + "/pmc_summary.json"))
b = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
run = json.load(open(src + "/run.json"))
# the PMC passes run the one-stream bench with --warmup 0: the solver kernels are launched for the attribution-free
# timed region only, so a kernel's per-launch average is over exactly the iterations the bench line covers
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps %d --warmup 0 --no-overlap`, "
                 "FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md), %s" % (run["steps"], name),
       "batch_per_gpu": b["config"]["batch_per_gpu"], "T": b["config"]["T"], "steps": run["steps"],
       "kernels": {k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "launches": v["launches"]} for k, v in p.items()},
       "linesearch_bytes_per_iteration": sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in p.items() if k.startswith("k_ls_")) / run["steps"]}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
