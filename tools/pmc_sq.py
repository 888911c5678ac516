import csv, glob, sys, collections
d = sys.argv[1]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("aoc64::", "").replace("aoc32::", "f32::")
    res[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in res.items():
    if not k.startswith("k_") or k.startswith("k_ls_plan") or k.startswith("k_ls_res") or k.startswith("k_ls_init"):
        continue
    avg = {c: sum(v) / len(v) for c, v in dd.items()}
    print(k, "launches", len(next(iter(dd.values()))))
    for c, v in sorted(avg.items()):
        print("    %-24s %.4g" % (c, v))
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in avg:
                print("      %s / WAVE_CYCLES = %.3f" % (c, avg[c] / wc))
    if "SQ_INSTS_VALU" in avg and "SQ_WAVES" in avg:
        print("      VALU insts per wave = %.0f ; per wave-stage = %.1f" % (avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"], avg["SQ_INSTS_VALU"] / avg["SQ_WAVES"] / 499))
