"""How often does the horizon cut (another order of the same arithmetic, gains equal to ~1e-14) flip a float32 state?
The states of 640 trajectories solved in a 640-trajectory launch (cut) against the same in a 66 000 launch (sequential)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from aircraftoptimalcontrol_amd import batch as aoc
from aircraftoptimalcontrol_amd import problems
pr = problems.step_maneuver(1.0, 2e-3)
bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
out = []
for B, first in ((66000, 0), (640, 0), (640, 65000)):
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(problems.random_x0(B, seed=20260403, first=first))
    h = s.run_fixed(3)
    out.append(s.current())
(xb, ub), (xs, us), (xt, ut) = out
for x2, u2, lo in ((xs, us, 0), (xt, ut, 65000)):
    ref = xb[lo:lo + 640]
    d = ref != x2
    ulp = np.abs(ref.astype(np.float32).view(np.int32).astype(np.int64) - x2.astype(np.float32).view(np.int32).astype(np.int64))
    print("first %5d: states differing %d of %d (trajectories %d), max float32 ulps %d, inputs rel %.2e" % (
        lo, d.sum(), d.size, d.any(axis=(1, 2)).sum(), ulp.max(),
        (np.abs(ub[lo:lo + 640] - u2).max(2) / np.maximum(np.abs(ub[lo:lo + 640]).max(2), 1e-3)).max()))
