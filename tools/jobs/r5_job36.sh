# vector instructions per launch of every kernel of the headline iteration (one stream, 20 iterations): the issue-slot
# accounting of EXPERIMENTS.md "Two streams, the search as a token" from counters instead of from instruction counts
set -o pipefail
OUT=gpurun_out/r05sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -- python bench.py --steps 20 --warmup 0 --no-overlap --no-cpu-baseline --no-secondary --placement-candidates 1 > $OUT/pmc.log 2>&1 || { tail -5 $OUT/pmc.log; exit 1; }
python tools/pmc_sq.py $OUT/pmc > $OUT/sq_counters.txt
python - <<'PY' >> $OUT/sq_counters.txt
import csv, glob, collections
f = glob.glob("gpurun_out/r05sq/pmc/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aoc64::", "")
    if r["Counter_Name"] == "SQ_INSTS_VALU": tot[k] += float(r["Counter_Value"]); n[k] += 1
RATE = 1024 * 2.4e9 / 4
print("\nSQ_INSTS_VALU per launch, and as ms of the device's issue rate (1024 SIMDs, one vector instruction of a wavefront")
print("per 4 cycles, 2.4 GHz: 6.14e11 per second).  An iteration = 1 backward + 1 forward + 1 k_ls_final + 2 + 2 trial launches:")
it = 0.0
nf = n["k_forward<true, false, 2, true, float>"]
for k, v in sorted(tot.items(), key=lambda kv: -kv[1]):
    if v / sum(tot.values()) > 1e-3:
        per_it = v / nf if k.startswith(("k_ls", "k_forward", "k_backward")) else 0.0
        it += per_it
        print("  %-44s launches %4d  %.4g per launch = %.3f ms of issue" % (k, n[k], v / n[k], v / n[k] / RATE * 1e3))
print("  per iteration (averaged over the %d iterations of the run): %.4g = %.2f ms of issue at 2.4 GHz" % (nf, it, it / RATE * 1e3))
PY
cat $OUT/sq_counters.txt | tail -25
