"""Converge mode, located: per-iteration timeline of aoc_newton_solve (aoc_solve_trace: batch in flight, ms) beside the
number of trajectories really iterating (from the histories of a second, identical solve) -> where the time goes:
lanes of stopped trajectories riding along, small generations, the tail.
    python tools/converge_timeline.py B [sync_every]"""
import ctypes as C
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems
from aircraftoptimalcontrol_amd._lib import lib


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    sync = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = torch.from_numpy(problems.perturbed_x0(pr, B, seed=20260401)).cuda()
    prm = aoc.make_params(max_iters=60, stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    r = s.solve_on_device(sync_every=sync, history=True, to_host=False)       # histories: who iterates when
    active = np.isfinite(r["history"]["descent"]).sum(0) if r["history"]["descent"].size else np.zeros(0)
    ntr = r["history"]["ntrials"]
    rows = np.zeros((256, 6))
    best = None
    for rep in range(3):
        s.set_initial_from_x0(x0)
        lib().aoc_solve_trace(rows.ctypes.data_as(C.c_void_p), rows.shape[0])
        r2 = s.solve_on_device(sync_every=sync, history=False, to_host=False)
        n = lib().aoc_solve_trace_rows()
        lib().aoc_solve_trace(None, 0)
        tl = rows[:n].copy()
        if best is None or tl[:, 5].max() < best[:, 5].max():
            best = tl
    tl = best
    it = r["iters"]
    total = float(tl[:, 5].max())
    out = {"B": B, "T": pr.T, "sync_every": sync, "device_ms": total, "device_s_wall": r2["device_seconds"],
           "iters_sum": int(it.sum()), "trajectory_iterations_per_s": float(it.sum() / total * 1e3),
           "iteration_hbm_frac": float(it.sum() / total * 1e3 * 496 * pr.T / 8e12),
           "iters_hist": {int(k): int(v) for k, v in zip(*np.unique(it, return_counts=True))}}
    print(json.dumps(out))
    lanes = 0
    for part in sorted(set(tl[:, 0].astype(int))):
        rows_p = tl[tl[:, 0] == part]
        dt = np.diff(np.concatenate([[0.0], rows_p[:, 5]]))
        print("part %d\nkk  in_flight tiles  read_left  ms_iter  ms_cum | iterating (whole batch)  mean_trials" % part)
        for i in range(len(rows_p)):
            kk = int(rows_p[i, 1])
            a = int(active[kk]) if 0 <= kk < len(active) else 0
            tr = float(ntr[:, kk][ntr[:, kk] >= 0].mean()) if 0 <= kk < ntr.shape[1] and a else 0.0
            print("%2d %9d %5d %9d %8.3f %8.2f | %9d %6.2f" % (kk, rows_p[i, 2], rows_p[i, 3], rows_p[i, 4], dt[i], rows_p[i, 5], a, tr))
        lanes += int(rows_p[:-1, 2].sum())
    print("sum over iterations of (in flight) = %d lane-iterations for %d useful ones" % (lanes, int(it.sum())))


if __name__ == "__main__":
    main()
