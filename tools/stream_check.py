"""aoc_streams_concurrent against the measured pair table: eight streams, the verdict for every pair, and the iteration time
of one two-stream solver on a few pairs of each kind; then batch.concurrent_streams and six solvers built with it.
    python tools/stream_check.py"""
import ctypes as C
import itertools
import sys
import time

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems
from aircraftoptimalcontrol_amd._lib import lib


def main():
    B = 131072
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    streams = [torch.cuda.Stream() for _ in range(8)]
    ok = {}
    for a, b in itertools.combinations(range(8), 2):
        ok[(a, b)] = lib().aoc_streams_concurrent(C.c_void_p(streams[a].cuda_stream), C.c_void_p(streams[b].cuda_stream))
    print("pairs that share a queue:", [p for p, v in ok.items() if v != 1])
    s2 = aoc.TwoStreamNewtonSolver(bp, B, prm, streams=streams[:2])

    def run(sv):
        best = None
        for rep in range(2):
            sv.set_initial_from_x0(x0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(10):
                sv.iterate(k)
            sv.join()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
            best = dt if best is None else min(best, dt)
        return best
    bad = [p for p, v in ok.items() if v != 1][:3]
    good = [p for p, v in ok.items() if v == 1][:3]
    for kind, pairs in (("shared queue", bad), ("concurrent", good)):
        for a, b in pairs:
            s2.streams = [streams[a], streams[b]]
            print("  %s (%d, %d): %.3f ms per iteration" % (kind, a, b, run(s2)), flush=True)
    del s2
    torch.cuda.empty_cache()
    solvers = [aoc.TwoStreamNewtonSolver(bp, B, prm) for _ in range(6)]      # streams through batch.concurrent_streams
    print("six solvers with checked streams: %s ms per iteration" % [round(run(s), 3) for s in solvers], flush=True)


if __name__ == "__main__":
    main()
