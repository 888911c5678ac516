"""Placement or streams?  N two-stream solvers built one after the other in one process, six-iteration probe and a
20-iteration run on each; AOC_SHARED_STREAMS=1: all of them on the same two streams (the always-slow second solver
disappears: its own two streams shared a hardware queue, EXPERIMENTS.md round 4).  (The run recorded there with
AOC_ARENA=1 — every solver's big buffers as views into ONE allocation — used a branch of batch.py that was removed again:
same pattern as with seven allocations.)     [AOC_SHARED_STREAMS=1] python tools/arena_lottery.py [N=6]"""
import os
import sys
import time

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    B, N = 131072, int(sys.argv[1]) if len(sys.argv) > 1 else 6
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    shared = [torch.cuda.Stream(), torch.cuda.Stream()] if os.environ.get("AOC_SHARED_STREAMS", "0") == "1" else None
    solvers = [aoc.TwoStreamNewtonSolver(bp, B, prm, streams=shared) for _ in range(N)]
    out = []
    for i, s2 in enumerate(solvers):
        row = []
        for Kk in (6, 6, 20):
            s2.set_initial_from_x0(x0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(Kk):
                s2.iterate(k)
            s2.join()
            torch.cuda.synchronize(); row.append((time.perf_counter() - t0) / Kk * 1e3)
        out.append((round(min(row[:2]), 3), round(row[2], 3)))
    print("shared streams %s " % os.environ.get("AOC_SHARED_STREAMS", "0") + "AOC_ARENA=%s: (6-iteration probe, 20-iteration run) ms per iteration by solver: %s" % (os.environ.get("AOC_ARENA", "0"), out), flush=True)


if __name__ == "__main__":
    main()
