"""Converge mode (aoc_newton_solve2): device milliseconds by re-packing threshold, sync interval and stream split, ONE
solver (one allocation) for every setting.   python tools/converge_sweep.py B"""
import json
import sys

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems
from aircraftoptimalcontrol_amd._lib import tuning


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = torch.from_numpy(problems.perturbed_x0(pr, B, seed=20260401)).cuda()
    prm = aoc.make_params(max_iters=60, stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)

    def run(sync=4, two=True, **kw):
        best = None
        with tuning(**kw):
            for rep in range(3):
                s.set_initial_from_x0(x0)
                r = s.solve_on_device(sync_every=sync, history=False, to_host=False, two_streams=two)
                best = r["device_seconds"] if best is None else min(best, r["device_seconds"])
        return round(best * 1e3, 2), int(r["iters"].sum())

    base, n = run()
    print(json.dumps({"B": B, "default_ms": base, "trajectory_iterations": n, "per_s": n / base * 1e3}), flush=True)
    for pct in (50, 60, 70, 80, 90, 97):
        for fast in (1, 2, 3):
            print("repack_pct %3d sync_fast %d: %7.2f ms" % (pct, fast, run(solve_repack_pct=pct, solve_sync_fast=fast)[0]), flush=True)
    print("one stream: %.2f ms" % run(two=False)[0])
    nt = (B + 63) // 64
    print("split at %d tiles (two halves of %d): %.2f ms" % (nt, nt // 2, run(solve_split_tiles=nt)[0]), flush=True)
    print("... with one-wavefront kernels for the halves (split_tiles=split_bw_tiles=%d): %.2f ms"
          % (nt // 2 - 1, run(solve_split_tiles=nt, split_tiles=nt // 2 - 1, split_bw_tiles=nt // 2 - 1)[0]), flush=True)
    print("sync_every 2: %.2f, 8: %.2f" % (run(sync=2)[0], run(sync=8)[0]))


if __name__ == "__main__":
    main()
