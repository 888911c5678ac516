"""Per-trajectory reference curves (SURVEY 8b `aoc_problem`: "xx_ref ptr + stride (shared or per-trajectory)"):
NewtonMethod captures xx_ref/uu_ref per INSTANCE (optcon.py:335-339), so a batch may mix manoeuvres and MPC instances
may track different targets.  aoc_problem.ref_per_traj = 1: `ref` is a tiled C = 8 array, one curve per trajectory."""
import numpy as np
import pytest

from conftest import rel_err
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _two_manoeuvres(B, T=500):
    """Trajectory b < B/2 climbs 2.71 m over 16 m (main_newton_method.py's step), the others 1.2 m over 14 m with
    another equilibrium thrust in uu_ref."""
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(1.0, 1.0 / T)
    XR = np.repeat(pr.xx_ref[None], B, 0); UR = np.repeat(pr.uu_ref[None], B, 0)
    h = B // 2
    XR[h:, 0] *= 14.0 / 16.0
    XR[h:, 1] *= 1.2 / 2.71
    XR[h:, 2] = np.sqrt(np.gradient(XR[h, 1], pr.tt) ** 2 + 14.0 ** 2)[None]
    UR[h:, 0] = 40.0
    return pr, XR, UR


@pytest.mark.parametrize("small_batch_kernels", [True, False])
def test_two_manoeuvres_in_one_batch_vs_oracle(small_batch_kernels, tuned):
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    if not small_batch_kernels:   # the one-wavefront-per-tile kernels and the work-list line search of large batches
        tuned(split_tiles=0, split_bw_tiles=0, nspec=2, ls_worklist=1)
    B = 150
    pr, XR, UR = _two_manoeuvres(B)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, XR, UR, pr.dt)
    assert bp.per_traj
    x0 = problems.perturbed_x0(pr, B, seed=12)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    oprm = orc.params()
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)                       # P-controller guess against each trajectory's own curve
    sample = [0, 1, 63, 64, 74, 75, 76, 128, 149]
    ops = {b: orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, XR[b], UR[b], pr.dt) for b in sample}
    xi, ui = s.current()
    mdl = orc.default_model(pr.dt)
    for b in sample:
        xr = XR[b].copy(); xr[:, 0] = x0[b]
        xo, uo = orc.initial_trajectory(mdl, xr)
        assert np.array_equal(xi[b], xo) and np.array_equal(ui[b], uo)
    for kk in range(11):                            # across the Hessian switch
        xi, ui = s.current()
        s.iterate(kk)
        sc = s.scalars()
        xn, un = s.current()
        for b in sample:
            r = orc.newton_iterate(ops[b], oprm, kk, xi[b], ui[b], xi[b][:, 0])
            assert r["stepsize"] == sc["stepsize"][b] and r["ntrials"] == sc["ntrials"][b], (kk, b)
            assert abs(r["J"] - sc["cost"][b]) <= 1e-12 * abs(r["J"])
            assert abs(r["descent"] - sc["descent"][b]) <= 1e-8 * abs(r["descent"])
            assert rel_err(un[b], r["uu"], 1e-3) < 1e-8
            assert np.array_equal(xn[b], r["xx"]) or rel_err(xn[b], r["xx"], 1e-2) < 5e-6
    # the two halves really solve different problems
    assert abs(sc["cost"][0] - sc["cost"][B - 1]) > 1e-3 * sc["cost"][0]


def test_per_trajectory_copies_of_one_curve_equal_the_shared_curve(tuned):
    """B copies of the shared curve as per-trajectory curves: every entry point that reads the reference gives the
    results of the shared form bit for bit (Newton iterations on both kernel families, gradient iteration, stored and
    rolled-out costs, initial guess, device solve)."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    B = 200
    x0 = problems.perturbed_x0(pr, B, seed=4)
    shared = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    per = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, np.repeat(pr.xx_ref[None], B, 0), np.repeat(pr.uu_ref[None], B, 0), pr.dt)
    prm = aoc.make_params(max_iters=30, stepsize_0=1.0, armijo_maxiters=10)
    for knobs in (dict(), dict(split_tiles=0, split_bw_tiles=0, nspec=2, ls_worklist=1), dict(nspec=3, ls_worklist=0, split_tiles=0, trial_split=0)):
        tuned(**knobs)
        out = []
        for bp in (shared, per):
            s = aoc.NewtonBatchSolver(bp, B, prm)
            s.set_initial_from_x0(x0)
            h = s.run_fixed(11)
            out.append((h, s.current()))
        (ha, (xa, ua)), (hb, (xb, ub)) = out
        assert np.array_equal(xa, xb) and np.array_equal(ua, ub), knobs
        for a, b in zip(ha, hb):
            for key in a:
                assert np.array_equal(a[key], b[key], equal_nan=True), (knobs, key)
    tuned()
    res = []
    for bp in (shared, per):
        g = aoc.GradientBatchSolver(bp, B, aoc.make_params(stepsize_0=1e-1, armijo_maxiters=20))
        g.set_initial_from_x0(x0)
        g.iterate(0)
        s = aoc.NewtonBatchSolver(bp, B, prm)
        s.set_initial_from_x0(x0)
        dev = s.solve_on_device(sync_every=3)
        xx, uu = g.current()
        res.append((g.scalars(), g.direction(), aoc.traj_cost(bp, xx, uu), aoc.rollout_cost(bp, x0, uu, 0.3 * uu, np.full(B, 0.5)),
                    dev["xx_star"], dev["uu_star"], dev["iters"]))
    a, b = res
    for k in a[0]:
        assert np.array_equal(a[0][k], b[0][k]), k
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    for u, v in zip(a[3], b[3]):
        assert np.array_equal(u, v)
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])


def test_receding_horizon_with_per_instance_targets():
    """MPC instances tracking different targets: the loop of tests/test_gpu_mpc.py with one long reference curve per
    instance (a window of the tiled curve array per step: ref_T + pointer offset), against the oracle per instance."""
    from aircraftoptimalcontrol_amd import mpc, problems
    T, L = 120, 170
    full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
    B, n_newton, n_steps, cold = 5, 2, 3, 5
    scale = np.linspace(0.6, 1.4, B)
    XR = np.repeat(full.xx_ref[None], B, 0); UR = np.repeat(full.uu_ref[None], B, 0)
    XR[:, 1] *= scale[:, None]                      # every instance climbs to its own height
    pr = problems.ProblemData("mpc-per-instance", full.QQt, full.RRt, full.QQT, XR, UR, full.tt, full.tf, full.dt)
    tw = problems.tracking_weights()
    rh = mpc.RecedingHorizon(pr, tw, B, T, n_newton=n_newton, sigma=None, seed=7, horizon_steps=64)
    x0 = XR[:, :, 0] + np.random.default_rng(1).normal(0, 1, (B, 6)) * problems.SIGMA_X0
    rh.start(x0, cold_iters=cold)
    mdl = orc.default_model(pr.dt)
    oprm = orc.params()
    oprob = lambda b, s: orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, mpc.window(XR[b], s, T), mpc.window(UR[b], s, T), pr.dt)
    XX, UU = [], []
    for b in range(B):
        xr = mpc.window(XR[b], 0, T).copy(); xr[:, 0] = x0[b]
        xx, uu = orc.initial_trajectory(mdl, xr)
        for kk in range(cold):
            r = orc.newton_iterate(oprob(b, 0), oprm, kk, xx, uu, x0[b])
            xx, uu = r["xx"], r["uu"]
        XX.append(xx); UU.append(uu)
    xg, ug = rh.solver.current()
    for b in range(B):
        assert np.array_equal(xg[b], XX[b]) and rel_err(ug[b], UU[b], 1e-3) < 1e-8
    x_true = x0.copy()
    for s in range(n_steps):
        out = rh.step()
        for b in range(B):
            op = oprob(b, s + 1)
            _, _, KK, _ = orc.lqr_tracking(mdl, tw[0], tw[1], tw[2], XX[b], UU[b], np.zeros(6))
            u_cl = UU[b][:, 0] + KK[:, :, 0] @ (x_true[b] - XX[b][:, 0])
            assert rel_err(out["u_applied"][b], u_cl, 1e-3) < 1e-8
            xn = orc.step(mdl, x_true[b], u_cl)[0]
            assert np.array_equal(out["x_true"][b], xn), (s, b)
            us = UU[b].copy(); us[:, :T - 2] = UU[b][:, 1:T - 1]
            xx, uu = orc.get_update(op, 0.0, us, np.zeros_like(us), xn)
            for kk in range(n_newton):
                r = orc.newton_iterate(op, oprm, kk, xx, uu, xn)
                xx, uu = r["xx"], r["uu"]
            XX[b], UU[b], x_true[b] = xx, uu, xn
        xg, ug = rh.solver.current()
        for b in range(B):
            assert np.array_equal(xg[b], XX[b]), (s, b)
            assert rel_err(ug[b], UU[b], 1e-3) < 1e-8, (s, b)
    assert np.ptp([orc.traj_cost(oprob(b, n_steps), XX[b], UU[b]) for b in range(B)]) > 0
