#!/usr/bin/env python3
"""Batched solve: B perturbed copies of the step maneuver to convergence (reference termination and
return-index semantics per trajectory), summary on stdout.

    python examples/run_batch.py --batch 4096 --horizon 500
"""
import argparse

import numpy as np

import _common  # noqa: F401
from aircraftoptimalcontrol_amd import batch, problems


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=500)
    ap.add_argument("--max-iters", type=int, default=60)
    ap.add_argument("--host-loop", action="store_true", help="iterate from Python (solve) instead of aoc_newton_solve")
    a = ap.parse_args()
    pr = problems.step_maneuver(1.0, 1.0 / a.horizon)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    s = batch.NewtonBatchSolver(bp, a.batch, batch.make_params(max_iters=a.max_iters, stepsize_0=1.0, armijo_maxiters=10))
    s.set_initial_from_x0(problems.perturbed_x0(pr, a.batch))
    r = s.solve() if a.host_loop else s.solve_on_device()
    h = r["history"]
    print("trajectories %d  converged %d  iterations min/median/max %d/%d/%d" % (
        a.batch, int(r["converged"].sum()), r["iters"].min(), int(np.median(r["iters"])), r["iters"].max()))
    print("cost first/last (mean): %.6g -> %.6g" % (h["cost"][:, 0].mean(), np.nanmean(h["cost"][np.arange(a.batch), r["iters"] - 1])))


if __name__ == "__main__":
    main()
