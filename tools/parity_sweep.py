"""Wide parity sweep (checker = the C oracle): every iteration of a batch is teacher-forced — the oracle
redoes ONE iteration from the GPU's own iterate, and the accepted step, the trial count, the costs, the new
inputs and the new states are compared per trajectory.  Writes one JSON line (kept under profiles/).

    python tools/parity_sweep.py [B=4096] [iters=12] [dist=random|perturbed] [problem=step|acro]
"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from aircraftoptimalcontrol_amd import batch as aoc, problems
from oracle import oracle as orc


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    n_it = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    dist = sys.argv[3] if len(sys.argv) > 3 else "random"
    prob = sys.argv[4] if len(sys.argv) > 4 else "step"
    pr = problems.step_maneuver(1.0, 2e-3) if prob == "step" else problems.acrobatic(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.random_x0(B, seed=4242) if dist == "random" else problems.perturbed_x0(pr, B, seed=4242)
    if prob == "acro" and dist == "random":
        x0[:, 2] = np.clip(x0[:, 2], 9.0, 12.0)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    oprm = orc.params(stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    rows = []
    t_or = 0.0
    for kk in range(n_it):
        xi, ui = s.current()
        s.iterate(kk)
        sc = s.scalars()
        xn, un = s.current()
        X, U = np.ascontiguousarray(xi), np.ascontiguousarray(ui)
        t0 = time.time()
        r = orc.newton_iterate_batch(op, oprm, X, U, xi[:, :, 0].copy(), kk, 1)
        t_or += time.time() - t0
        fin = np.isfinite(sc["cost_new"]) & np.isfinite(r["cost"][:, 0])
        Jg_new = sc["cost_new"]
        Jo_new = np.array([np.nan] * B)
        # cost of the oracle's new iterate
        for b in range(0, B, max(B // 256, 1)):
            Jo_new[b] = orc.traj_cost(op, X[b], U[b])
        smp = np.isfinite(Jo_new) & np.isfinite(Jg_new)
        step_eq = (r["stepsize"][:, 0] == sc["stepsize"]) | ~fin
        ntr_eq = (r["ntrials"][:, 0] == sc["ntrials"]) | ~fin
        den = np.maximum(np.abs(U), 1e-3)
        both = np.isfinite(un) & np.isfinite(U)
        uerr = np.where(both, np.abs(un - U) / den, 0.0).max(axis=(1, 2))      # elementwise, floor 1e-3 (SURVEY 8c gate)
        # the same differences against the size of the input channel: max_t |du_c| / max_t |u_c|
        usc = (np.where(both, np.abs(un - U), 0.0).max(axis=2) / np.maximum(np.abs(np.where(both, U, 0.0)).max(axis=2), 1e-3)).max(axis=1)
        flagged = (sc["status"] & (4 | 8)) != 0       # singular / regularised gains in this iteration
        s.status.zero_()
        x_same = np.array([np.array_equal(xn[b], X[b], equal_nan=True) for b in range(B)])
        ok = step_eq & ntr_eq                      # same Armijo verdicts: the iterates are comparable
        rows.append(dict(kk=kk, finite=int(fin.sum()),
                         step_mismatch=int((~step_eq).sum()), ntrials_mismatch=int((~ntr_eq).sum()),
                         cost_rel_max=float(np.max(np.abs(r["cost"][:, 0] - sc["cost"])[fin] / np.abs(sc["cost"][fin]))),
                         descent_rel_max=float(np.max(np.abs(r["descent"][:, 0] - sc["descent"])[fin] / np.abs(sc["descent"][fin]))),
                         new_cost_rel_max_sampled=float(np.max(np.abs(Jo_new - Jg_new)[smp & ok] / np.abs(Jg_new[smp & ok]))) if (smp & ok).any() else None,
                         u_rel_max=float(uerr[ok & fin].max()), u_rel_median=float(np.median(uerr[ok & fin])),
                         n_over_1e8=int((uerr[ok & fin] > 1e-8).sum()),
                         u_channel_rel_max=float(usc[ok & fin].max()), n_regularised=int(flagged.sum()),
                         u_rel_max_unflagged=float(uerr[ok & fin & ~flagged].max()),
                         u_channel_rel_max_unflagged=float(usc[ok & fin & ~flagged].max()),
                         x_bit_identical=int((x_same & ok & fin).sum()), comparable=int((ok & fin).sum())))
        print(json.dumps(rows[-1]), file=sys.stderr)
    out = dict(B=B, T=pr.T, iterations=n_it, x0=dist, problem=prob, oracle_seconds=round(t_or, 1), per_iteration=rows,
               worst_u_rel=max(r_["u_rel_max"] for r_ in rows),
               worst_u_channel_rel_unflagged=max(r_["u_channel_rel_max_unflagged"] for r_ in rows),
               total_step_mismatch=sum(r_["step_mismatch"] for r_ in rows),
               total_x_not_identical=sum(r_["comparable"] - r_["x_bit_identical"] for r_ in rows))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
