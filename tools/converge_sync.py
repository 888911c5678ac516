"""Converge mode (aoc_newton_solve): device seconds by how often the host reads the count of still-iterating trajectories
(sync_every), for 4096 and 65 536 perturbed step-maneuver trajectories, max_iters 60."""
import sys

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(max_iters=60, stepsize_0=1.0, armijo_maxiters=10)
    for B in (4096, 65536):
        s = aoc.NewtonBatchSolver(bp, B, prm)
        x0 = torch.from_numpy(problems.perturbed_x0(pr, B, seed=20260401)).cuda()
        for sync in (1, 2, 3, 4, 6, 8):
            best = None
            for rep in range(3):
                s.set_initial_from_x0(x0)
                r = s.solve_on_device(sync_every=sync, history=False, to_host=False)
                best = r["device_seconds"] if best is None else min(best, r["device_seconds"])
            print("B %6d  sync_every %d  %.4f s  (%.3g solved trajectories/s, last_kk %d)" % (B, sync, best, B / best, r["last_kk"]), flush=True)


if __name__ == "__main__":
    main()
