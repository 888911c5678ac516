"""Receding-horizon loop (BASELINE configs[4]) on the HIP library against the same loop on the oracle,
re-solve by re-solve: gains, closed-loop input, plant state, warm-started re-solve."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_receding_horizon_vs_oracle():
    from aircraftoptimalcontrol_amd import mpc, problems
    T, L = 200, 260
    full = problems.step_maneuver(tf=1.0, dt=1.0 / L)      # a long reference curve to slide over
    pr = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
    tw = problems.tracking_weights()
    B, n_newton, n_steps, cold = 6, 2, 4, 6
    sigma = np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002])
    rh = mpc.RecedingHorizon(pr, tw, B, T, n_newton=n_newton, sigma=sigma, seed=7)
    x0 = problems.perturbed_x0(pr, B, seed=3)
    rh.start(x0, cold_iters=cold)

    # ---- the same loop on the oracle, one instance at a time
    mdl = orc.default_model(pr.dt)
    oprm = orc.params()

    def oprob(s):
        return orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, mpc.window(pr.xx_ref, s, T), mpc.window(pr.uu_ref, s, T), pr.dt)

    XX, UU = [], []
    op = oprob(0)
    for b in range(B):
        xr = mpc.window(pr.xx_ref, 0, T).copy(); xr[:, 0] = x0[b]
        xx, uu = orc.initial_trajectory(mdl, xr)
        for kk in range(cold):
            r = orc.newton_iterate(op, oprm, kk, xx, uu, x0[b])
            xx, uu = r["xx"], r["uu"]
        XX.append(xx); UU.append(uu)
    xg, ug = rh.solver.current()
    for b in range(B):
        assert np.array_equal(xg[b], XX[b]) and rel_err(ug[b], UU[b], 1e-3) < 1e-8
    x_true = x0.copy()
    for s in range(n_steps):
        out = rh.step()
        dist = rh.disturbance(s)
        op = oprob(s + 1)
        for b in range(B):
            _, _, KK, ns = orc.lqr_tracking(mdl, tw[0], tw[1], tw[2], XX[b], UU[b], np.zeros(6))
            K0 = KK[:, :, 0]
            assert np.max(np.abs(out["K0"][b] - K0) / np.abs(K0).max()) < 1e-8
            u_cl = UU[b][:, 0] + K0 @ (x_true[b] - XX[b][:, 0])
            assert rel_err(out["u_applied"][b], u_cl, 1e-3) < 1e-8
            xn = orc.step(mdl, x_true[b], u_cl)[0] + dist[b]
            # the plant state is float32-rounded by step(): bit-identical unless u differs at a tie
            assert np.array_equal(out["x_true"][b], xn), (s, b)
            us = UU[b].copy(); us[:, :T - 2] = UU[b][:, 1:T - 1]
            xx, uu = orc.get_update(op, 0.0, us, np.zeros_like(us), xn)
            for kk in range(n_newton):
                r = orc.newton_iterate(op, oprm, kk, xx, uu, xn)
                xx, uu = r["xx"], r["uu"]
            XX[b], UU[b], x_true[b] = xx, uu, xn
            assert abs(out["cost"][b] - orc.traj_cost(op, xx, uu)) <= 1e-9 * abs(out["cost"][b])
        xg, ug = rh.solver.current()
        for b in range(B):
            assert np.array_equal(xg[b], XX[b]), (s, b)
            assert rel_err(ug[b], UU[b], 1e-3) < 1e-8, (s, b)
