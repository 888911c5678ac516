"""Converge-mode timing (SURVEY 8d config 2 "converge-mode run", scaled up): every trajectory iterates until its
descent test stops it.  Compares the host loop with re-packing, the host loop without, and aoc_newton_solve."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.perturbed_x0(pr, B, seed=20260401)
    prm = aoc.make_params(max_iters=60, stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    out = {"B": B, "T": pr.T, "max_iters": 60}
    ref = None
    for name, fn in (("host_loop_repacking", lambda: s.solve()),
                     ("host_loop", lambda: s.solve(compact=False)),
                     ("device_solve_sync4", lambda: s.solve_on_device(sync_every=4)),
                     ("device_solve_sync4_nohist", lambda: s.solve_on_device(sync_every=4, history=False))):
        best = None
        for rep in range(2):
            s.set_initial_from_x0(x0)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        out[name + "_s"] = round(best, 4)
        if "device_seconds" in r:
            out[name + "_device_only_s"] = round(r["device_seconds"], 4)
        if ref is None:
            ref = r
            out["iters_mean"] = float(r["iters"].mean()); out["iters_max"] = int(r["iters"].max())
            out["converged"] = int(r["converged"].sum())
        else:
            out[name + "_identical"] = bool(np.array_equal(r["xx_star"], ref["xx_star"], equal_nan=True) and
                                            np.array_equal(r["uu_star"], ref["uu_star"], equal_nan=True) and
                                            np.array_equal(r["iters"], ref["iters"]))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
