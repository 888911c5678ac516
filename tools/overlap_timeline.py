"""Is the line-search search phase of one half batch hidden behind the streaming passes of the other?  From a rocprofv3
--kernel-trace of the two-stream bench run (profile_round.sh: trace2): kernels are attributed to a half by their grid
(the two halves are separate solvers, launched on two streams = two HSA queues), then for every search kernel
(k_ls_init_wl / k_ls_plan_wl / k_ls_trial_wl / k_ls_replan) of one queue the share of its duration during which the OTHER
queue runs a streaming pass (k_backward / k_forward / k_ls_final) is summed up, per iteration regime.
    python tools/overlap_timeline.py <trace dir> [out.json]"""
import collections
import csv
import glob
import json
import sys


def short(n):
    return n.split("(")[0].replace("void ", "").replace("aoc64::", "").split("<")[0]


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if "aoc64::k_" in r["Kernel_Name"]]
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey], short(r["Kernel_Name"])) for r in rows]
    ev.sort()
    queues = collections.Counter(q for _, _, q, n in ev if n in ("k_backward", "k_forward"))
    # the two queues that carry the halves of the LAST run (the timed region comes last among the two-stream runs)
    qs = [q for q, _ in queues.most_common(2)]
    if len(qs) < 2:
        print("one queue only: not a two-stream trace"); return
    stream = {q: [e for e in ev if e[2] == q] for q in qs}
    STREAMING = ("k_backward", "k_forward", "k_ls_final")
    SEARCH = ("k_ls_init_wl", "k_ls_plan_wl", "k_ls_trial_wl", "k_ls_replan")
    out = {"queues": qs, "per_queue": {}}
    for q in qs:
        other = [e for e in stream[[x for x in qs if x != q][0]]]
        o_stream = [(s, e) for s, e, _, n in other if n in STREAMING]
        o_any = [(s, e) for s, e, _, n in other]
        # iteration index of this queue: count of k_backward launches so far (the last K+W launches are the timed run)
        it, tot, hid, hid_any, by_it = -1, 0, 0, 0, collections.defaultdict(lambda: [0, 0])
        for s, e, _, n in stream[q]:
            if n == "k_backward":
                it += 1
            if n not in SEARCH:
                continue
            d = e - s
            ov = sum(max(0, min(e, b) - max(s, a)) for a, b in o_stream if b > s and a < e)
            ova = sum(max(0, min(e, b) - max(s, a)) for a, b in o_any if b > s and a < e)
            tot += d; hid += min(ov, d); hid_any += min(ova, d)
            by_it[it][0] += d; by_it[it][1] += min(ov, d)
        n_it = it + 1
        last = sorted(by_it)[-20:]
        out["per_queue"][q] = {
            "iterations_seen": n_it, "search_ms_total": tot / 1e6, "share_behind_a_streaming_pass_of_the_other_half": hid / max(tot, 1),
            "share_behind_any_kernel_of_the_other_half": hid_any / max(tot, 1),
            "last_20_iterations": [{"search_ms": round(by_it[i][0] / 1e6, 3), "hidden_share": round(by_it[i][1] / max(by_it[i][0], 1), 2)} for i in last]}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt)


if __name__ == "__main__":
    main()
