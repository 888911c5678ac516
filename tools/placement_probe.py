"""Does the duration of a streaming pass depend on WHERE its buffers were allocated?  The same 131 072-trajectory
solver is built several times in one process (fresh allocations each time, the earlier ones kept alive or freed) and
iterations 0..9 are timed pass by pass."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    keep = len(sys.argv) > 2 and sys.argv[2] == "keep"
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    alive = []
    for rep in range(6):
        s = aoc.NewtonBatchSolver(bp, B, prm)
        for r2 in range(2):
            s.set_initial_from_x0(x0)
            evs = [s.iterate_timed(k) for k in range(10)]
            torch.cuda.synchronize()
            ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in evs])
            print("alloc %d run %d  ws %#x xb %#x ub %#x  passes ms (kk 0..8 mean) %s   per-iteration backward %s" % (
                rep, r2, s.ws.data_ptr(), s.xb[0].data_ptr(), s.ub[0].data_ptr(), np.round(ms[:9].mean(0), 3),
                np.round(ms[:, 0], 2)), flush=True)
        if keep:
            alive.append(s)
        else:
            del s
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
