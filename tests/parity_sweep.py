"""Wide parity sweep, shared by tests/test_gpu_sweep.py and tools/parity_sweep.py (checker = the C oracle).

Every iteration of a batch is teacher-forced: the oracle redoes ONE iteration from the GPU's own iterate, and the
accepted step, the trial count, the costs, the descent, the new inputs and the new states are compared per
trajectory.  Error measures for the new inputs u' (B,2,T):
  channel-scaled   max_c  max_t |du_c| / max(max_t |u_c|, 1e-3)      — the size of the input channel is the scale
  elementwise      max_{c,t} |du| / max(|u|, 1e-3)                   — SURVEY 8c's gate; blows up at zero crossings of u
"""
import time

import numpy as np

from oracle import oracle as orc


def sweep(aoc, problems, B=4096, n_it=12, dist="random", prob="step", seed=4242, log=None, make_solver=None):
    """make_solver(bp, B, prm) -> the solver under test (default: one NewtonBatchSolver; a TwoStreamNewtonSolver has the
    same interface).  Which KERNELS it runs is decided by the batch size and aoc_tuning at the time of each iterate()."""
    # step maneuver at the bench's T = 500; the acrobatic problem at the reference's native T = 1000 (acrobatic_newton.py:72-76)
    pr = problems.step_maneuver(1.0, 2e-3) if prob == "step" else problems.acrobatic(1.0, 1e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.random_x0(B, seed=seed) if dist == "random" else problems.perturbed_x0(pr, B, seed=seed)
    if prob == "acro" and dist == "random":
        x0[:, 2] = np.clip(x0[:, 2], 9.0, 12.0)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    oprm = orc.params(stepsize_0=1.0, armijo_maxiters=10)
    s = (make_solver or aoc.NewtonBatchSolver)(bp, B, prm)
    s.set_initial_from_x0(x0)
    parts = getattr(s, "parts", [s])      # a two-stream solver keeps one status array per half
    rows = []
    t_or = 0.0
    for kk in range(n_it):
        xi, ui = s.current()
        s.iterate(kk)
        sc = s.scalars()
        xn, un = s.current()
        X, U = np.ascontiguousarray(xi), np.ascontiguousarray(ui)
        t0 = time.time()
        r = orc.newton_iterate_batch(op, oprm, X, U, xi[:, :, 0].copy(), kk, 1)     # X, U <- the oracle's new iterate
        t_or += time.time() - t0
        fin = np.isfinite(sc["cost_new"]) & np.isfinite(r["cost"][:, 0]) & np.isfinite(un).all((1, 2)) & np.isfinite(U).all((1, 2))
        step_eq = (r["stepsize"][:, 0] == sc["stepsize"]) | ~fin
        ntr_eq = (r["ntrials"][:, 0] == sc["ntrials"]) | ~fin
        d = np.where(np.isfinite(un) & np.isfinite(U), np.abs(un - U), 0.0)
        Uf = np.where(np.isfinite(U), U, 0.0)
        elem = (d / np.maximum(np.abs(Uf), 1e-3)).max(axis=(1, 2))
        chan = (d.max(axis=2) / np.maximum(np.abs(Uf).max(axis=2), 1e-3)).max(axis=1)
        flagged = (sc["status"] & (4 | 8)) != 0       # singular / regularised gains in this iteration
        reg_g = (sc["status"] & 8) != 0               # GPU: M + 0.5 I applied at some stage of this iteration (optcon.py:745-749)
        reg_o = r["nreg"][:, 0] > 0                   # oracle: the same, counted per stage
        s.join()
        for part in parts:
            part.status.zero_()
        x_same = np.array([np.array_equal(xn[b], X[b], equal_nan=True) for b in range(B)])
        ok = step_eq & ntr_eq & fin                   # same Armijo verdicts: the iterates are comparable
        mx = lambda a, m: float(a[m].max()) if m.any() else 0.0
        rows.append(dict(
            kk=kk, finite=int(fin.sum()), comparable=int(ok.sum()),
            step_mismatch=int((~step_eq).sum()), ntrials_mismatch=int((~ntr_eq).sum()),
            cost_rel_max=mx(np.abs(r["cost"][:, 0] - sc["cost"]) / np.abs(sc["cost"]), fin),
            descent_rel_max=mx(np.abs(r["descent"][:, 0] - sc["descent"]) / np.abs(sc["descent"]), fin & ~flagged),
            n_regularised=int(flagged.sum()),
            # the regularisation branch itself: which trajectories took it, on the device and in the oracle (finite ones)
            n_regularised_gpu=int((reg_g & fin).sum()), n_regularised_oracle=int((reg_o & fin).sum()),
            n_regularised_set_difference=int(((reg_g != reg_o) & fin).sum()),
            # who diverges in THIS iteration: a new iterate (states or inputs) that is not finite, on either side
            nonfinite_gpu=int((~(np.isfinite(xn).all((1, 2)) & np.isfinite(un).all((1, 2)))).sum()),
            nonfinite_oracle=int((~(np.isfinite(X).all((1, 2)) & np.isfinite(U).all((1, 2)))).sum()),
            nonfinite_set_difference=int(((np.isfinite(xn).all((1, 2)) & np.isfinite(un).all((1, 2))) !=
                                          (np.isfinite(X).all((1, 2)) & np.isfinite(U).all((1, 2)))).sum()),
            u_channel_rel_max=mx(chan, ok), u_channel_rel_max_unflagged=mx(chan, ok & ~flagged),
            # where the gains were regularised the recursion is not contractive (Q3): the distribution, not only its worst case
            u_channel_rel_p99_flagged=float(np.percentile(chan[ok & flagged], 99.0)) if (ok & flagged).any() else 0.0,
            u_channel_rel_median_flagged=float(np.median(chan[ok & flagged])) if (ok & flagged).any() else 0.0,
            n_flagged_over_1e3=int((chan[ok & flagged] > 1e-3).sum()),
            u_elementwise_rel_max_unflagged=mx(elem, ok & ~flagged), u_elementwise_rel_median=float(np.median(elem[ok])) if ok.any() else 0.0,
            u_elementwise_rel_p999=float(np.percentile(elem[ok], 99.9)) if ok.any() else 0.0,
            u_channel_rel_median=float(np.median(chan[ok])) if ok.any() else 0.0,
            n_elementwise_over_1e8=int((elem[ok & ~flagged] > 1e-8).sum()),
            x_bit_identical=int((x_same & ok).sum())))
        if log:
            log(rows[-1])
    return dict(B=B, T=pr.T, iterations=n_it, x0=dist, problem=prob, oracle_seconds=round(t_or, 1), per_iteration=rows)
