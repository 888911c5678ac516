"""Per-pass times of iterations kk = 0..N-1 in fixed-iteration mode (nobody is stopped): from kk = 13 on a third of the
near-converged trajectories reject every Armijo candidate in every iteration (as in the reference)."""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from aircraftoptimalcontrol_amd import batch as aoc, problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
N = int(sys.argv[2]) if len(sys.argv) > 2 else 22
pr = problems.step_maneuver(1.0, 2e-3)
bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
x0 = problems.random_x0(B, seed=20260403)
prm = aoc.make_params(max_iters=200, stepsize_0=1.0, armijo_maxiters=10)
s = aoc.NewtonBatchSolver(bp, B, prm)
s.set_initial_from_x0(x0)
tot = 0.0
for kk in range(N):
    s.status.zero_()
    ev = s.iterate_timed(kk)
    torch.cuda.synchronize()
    sc = s.scalars()
    st = sc["status"]
    ms = [round(ev[i].elapsed_time(ev[i + 1]), 3) for i in range(2)] + [round(ev[2].elapsed_time(ev[4]), 3)]
    tot += sum(ms)
    print(kk, "ms", ms, "exhausted", int((st & 16 != 0).sum()), "mean trials", round(float(sc["ntrials"].mean()), 2))
print("mean ms per iteration", round(tot / N, 3))
