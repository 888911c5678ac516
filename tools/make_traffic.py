"""profiles/<tag>_* from a gpurun_out/<tag> directory produced by tools/profile_round.sh, and
profiles/traffic.json (what bench.py reports as roofline.traffic)."""
import json, shutil, sys, os
tag = sys.argv[1]; name = sys.argv[2] if len(sys.argv) > 2 else tag
src = "gpurun_out/" + tag
os.makedirs("profiles", exist_ok=True)
shutil.copy(src + "/kernel_stats.csv", "profiles/%s_kernel_stats.csv" % name)
shutil.copy(src + "/pmc_summary.json", "profiles/%s_pmc_summary.json" % name)
shutil.copy(src + "/bench.json", "profiles/%s_bench.json" % name)
p = json.load(open(src + "/pmc_summary.json"))
b = json.loads(open(src + "/bench.json").read().strip().splitlines()[-1])
def find(prefix):
    ks = [k for k in p if k.startswith(prefix)]
    return max(ks, key=lambda k: p[k]["launches"]) if ks else None
steps = 3  # pmc passes run 3 timed iterations with no warmup
kb, kf = find("k_backward"), find("k_forward")
# every kernel of the line search (final pass, gathered and dense trial rounds, gather, bookkeeping), per iteration
ls = sum(v["hbm_bytes_per_launch"] * v["launches"] for k, v in p.items() if k.startswith("k_ls_")) / steps
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), FETCH_SIZE x2 (gfx950), %s" % name,
       "batch_per_gpu": b["config"]["batch_per_gpu"], "T": b["config"]["T"],
       "kernels": {"backward": {"kernel": kb, "hbm_bytes_per_launch": p[kb]["hbm_bytes_per_launch"]},
                   "forward": {"kernel": kf, "hbm_bytes_per_launch": p[kf]["hbm_bytes_per_launch"]},
                   "linesearch": {"kernel": "all k_ls_* launches of one iteration (final pass, trial rounds, gather)", "hbm_bytes_per_launch": ls}}}
json.dump(out, open("profiles/traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
