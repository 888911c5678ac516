"""Two half batches on two streams: free-running against the search token (TwoStreamNewtonSolver.iterate_phased), on ONE
solver (one set of buffers: the placement is the same for both), headline workload.
    python tools/phase_probe.py [B] [iterations] [repetitions]"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main(B=131072, n=20, reps=4):
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    s = aoc.TwoStreamNewtonSolver(bp, B, prm)
    res = {}
    for rep in range(reps):
        for mode in ("free", "phased"):
            s.phased = mode == "phased"
            s.set_initial_from_x0(x0)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(n):
                s.iterate(k)
            s.join(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n * 1e3
            res.setdefault(mode, []).append(dt)
            if rep == reps - 1:
                res[mode + "_scalars"] = s.scalars()
    for mode in ("free", "phased"):
        print("%-7s ms per iteration over kk 0..%d: %s" % (mode, n - 1, np.round(res[mode], 3)), flush=True)
    a, b = res["free_scalars"], res["phased_scalars"]
    print("results identical:", all(np.array_equal(a[k], b[k], equal_nan=True) for k in a))


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:]))
