"""Per-iteration distribution of the Armijo trial count for the bench workload at a given batch size: decides how many
candidates are worth storing in the forward pass (a tile is updated by copying only if all its 64 trajectories accepted
a stored candidate)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from aircraftoptimalcontrol_amd import batch, problems

def main(B=4096, iters=14, perturbed=0):
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.perturbed_x0(pr, B, seed=20260401) if perturbed else problems.random_x0(B, seed=20260403)
    s = batch.NewtonBatchSolver(bp, B, batch.make_params(stepsize_0=1.0, armijo_maxiters=10))
    s.set_initial_from_x0(x0)
    for kk in range(iters):
        s.iterate(kk)
        n = s.scalars()["ntrials"][:B]
        tiles = n.reshape(-1, 64).max(axis=1)
        print("kk %2d  trials mean %.2f  max %2d  share<=2 %.3f <=4 %.3f <=6 %.3f | tiles with max<=2 %.2f <=4 %.2f <=6 %.2f <=10 %.2f" % (
            kk, n.mean(), n.max(), (n <= 2).mean(), (n <= 4).mean(), (n <= 6).mean(),
            (tiles <= 2).mean(), (tiles <= 4).mean(), (tiles <= 6).mean(), (tiles <= 10).mean()))

if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
