"""Spread of the two-stream iteration time over solver instances of ONE process (each with its own allocations): the
measurement behind batch.best_placed.   python tools/two_stream_lottery.py [instances=5]"""
import sys
import time

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    B, K, N = 131072, 20, int(sys.argv[1]) if len(sys.argv) > 1 else 5
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    solvers = [aoc.TwoStreamNewtonSolver(bp, B, prm) for _ in range(N)]
    for rep in range(2):
        for i, s2 in enumerate(solvers):
            for Kk in (6, K):
                s2.set_initial_from_x0(x0)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for k in range(Kk):
                    s2.iterate(k)
                s2.join()
                torch.cuda.synchronize(); dt = time.perf_counter() - t0
                print("rep %d solver %d  K=%2d  %.3f ms per iteration" % (rep, i, Kk, dt / Kk * 1e3), flush=True)


if __name__ == "__main__":
    main()
