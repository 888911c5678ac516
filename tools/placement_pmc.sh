#!/bin/bash
# Counters of the backward pass per workspace allocation: tools/placement_probe2.py under rocprofv3 --pmc (two counter sets,
# two processes), k_backward dispatches grouped by candidate (six iterations each) next to the times the probe prints.
OUT=gpurun_out/placement_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SETS=("TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_WRITE_GMI_32B_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum")
for i in 0 1; do
  rocprofv3 --pmc ${SETS[$i]} --output-format csv -d $OUT/set$i -- python tools/placement_probe2.py 131072 500 ws > $OUT/set$i.log 2>&1 || { tail -5 $OUT/set$i.log; exit 1; }
  grep -E "^base|^ws cand" $OUT/set$i.log
  python - $OUT/set$i <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_backward" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
names = sorted({k for d in by.values() for k in d})
print("k_backward dispatches:", len(ids))
for g in range(0, len(ids), 6):
    grp = ids[g:g + 6][1:]           # skip iteration 0 of each run, as the probe's mean does
    if not grp: break
    print("group %2d  " % (g // 6) + "  ".join("%s %.4g" % (n.replace("_sum", ""), sum(by[i].get(n, 0) for i in grp) / len(grp)) for n in names))
PY
done
