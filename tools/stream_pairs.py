"""Placement or queues?  ONE two-stream solver (its buffers never change), eight HIP streams created one after the other;
the same 10 iterations timed on pairs of them.  The third and fourth stream a process creates turned out to be slow ones
(tools/arena_lottery.py: the second solver of a process is always the slowest — unless all solvers share two streams).
    python tools/stream_pairs.py"""
import sys
import time

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    B = 131072
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    streams = [torch.cuda.Stream() for _ in range(8)]
    s2 = aoc.TwoStreamNewtonSolver(bp, B, prm, streams=streams[:2])
    import itertools
    cur = torch.cuda.current_stream()
    hp = torch.cuda.Stream(priority=-1)
    named = {"default": cur, "hi-prio": hp}
    pairs = list(itertools.combinations(range(8), 2)) + [("default", 0), ("default", 3), ("hi-prio", 0), ("hi-prio", 3)]
    grid = {}
    for a, b in pairs:
        s2.streams = [named.get(a, streams[a] if isinstance(a, int) else None), streams[b]]
        best = None
        for rep in range(2):
            s2.set_initial_from_x0(x0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(10):
                s2.iterate(k)
            s2.join()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
            best = dt if best is None else min(best, dt)
        grid[(a, b)] = best
        if not isinstance(a, int):
            print("streams (%s, %d): %.3f ms per iteration (kk 0..9)" % (a, b, best), flush=True)
    print("ms per iteration (kk 0..9) by stream pair (row a, column b):")
    for a in range(8):
        print("  s%d " % a + " ".join("%6.3f" % grid[(a, b)] if (a, b) in grid else "   -  " for b in range(8)), flush=True)


if __name__ == "__main__":
    main()
