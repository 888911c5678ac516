"""GradientMethod.optimize (reference optcon.py:27-174) on the device.  PARITY UNPINNED — the reference's
own method raises TypeError (armijo_stepsize called with 8 of its 9 arguments, optcon.py:125 vs :204), so there is no
reference output to compare with: the HIP path (aoc_gradient + aoc_linesearch) is checked against the oracle's
statement-by-statement restatement of the same loop with the missing JP argument supplied, and through the properties
a steepest-descent iteration must have."""
import contextlib
import io
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden, rel_err
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_gradient_iteration_vs_oracle_restatement():
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 70
    x0 = problems.perturbed_x0(pr, B, seed=3)
    prm = aoc.make_params(stepsize_0=1e-1, armijo_maxiters=20)      # GradientMethod's own defaults are 1e-2 / 20
    oprm = orc.params(stepsize_0=1e-1, armijo_maxiters=20)
    s = aoc.GradientBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    costs = []
    for kk in range(4):
        xi, ui = s.current()
        s.iterate(kk)
        sc = s.scalars()
        du = s.direction()
        xn, un = s.current()
        costs.append(sc["cost"].copy())
        assert (sc["descent"] < 0).all()                              # the slope: -sum |du|^2
        assert np.allclose(-sc["descent"], (du ** 2).sum((1, 2)), rtol=1e-12)
        acc = sc["ntrials"] < 20
        assert (sc["cost_new"][acc] <= sc["cost"][acc] + 0.5 * sc["stepsize"][acc] * sc["descent"][acc]).all()
        for b in (0, 17, 69):
            r = orc.gradient_iterate(op, oprm, xi[b], ui[b], xi[b][:, 0])
            assert abs(r["J"] - sc["cost"][b]) <= 1e-13 * abs(r["J"])
            assert rel_err(du[b], r["du"], 1e-9) < 1e-10
            assert abs(-sc["descent"][b] - r["descent"]) <= 1e-12 * r["descent"]
            assert r["stepsize"] == sc["stepsize"][b] and r["ntrials"] == sc["ntrials"][b]
            assert rel_err(un[b], r["uu"], 1e-3) < 1e-10
            assert np.array_equal(xn[b], r["xx"]) or rel_err(xn[b], r["xx"], 1e-2) < 5e-6   # float32 flip at most
    assert (costs[-1] < costs[0]).all()


def test_gradient_is_the_derivative_of_the_cost():
    """du = -dJ/du: the cost along u + eps*du falls at the rate -sum |du|^2 (finite difference through the nonlinear
    rollout; float32 state rounding limits it to a few per cent)."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.perturbed_x0(pr, 64, seed=9)
    s = aoc.GradientBatchSolver(bp, 64, aoc.make_params(stepsize_0=1e-1, armijo_maxiters=20))
    s.set_initial_from_x0(x0)
    xi, ui = s.current()
    s.iterate(0)
    du, sc = s.direction(), s.scalars()
    eps = 1e-3
    _, _, J1, _ = aoc.rollout_cost(bp, x0, ui, du, np.full(64, eps), write=False)
    fd = (J1 - sc["cost"]) / eps
    assert np.allclose(fd, sc["descent"], rtol=0.05)


def test_dropin_gradient_method_runs_like_the_reference_would():
    sys.path.insert(0, os.path.join(ROOT, "aircraftoptimalcontrol_amd", "dropin"))
    import aircraft_simplified as air
    import optcon
    pg = load_golden("problem_step_T500")
    c = load_golden("g6_chain_step_T500")
    dyn = air.Dynamics(); dyn.dt = float(pg["dt"])
    GM = optcon.GradientMethod(dyn, air.Cost(pg["QQt"], pg["RRt"], pg["QQT"]), pg["xx_ref"], pg["uu_ref"], max_iters=8,
                               stepsize_0=1e-1, cc=0.5, beta=0.7, armijo_maxiters=20)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        xx_star, uu_star = GM.optimize(c["xx_init"], c["uu_init"], float(pg["tf"]), float(pg["dt"]))
    out = buf.getvalue()
    it = re.findall(r"Iter = (\d+)\t Descent = (\S+)\t Cost = (\S+)", out)
    assert out.startswith("-*-*-*-*-*-\n") and len(it) == 7 and "term =" not in out        # optcon.py:81, :143
    cost = np.array([float(v[2]) for v in it]); desc = np.array([float(v[1]) for v in it])
    assert (np.diff(cost) < 0).all() and (desc > 0).all()
    assert abs(cost[0] - c["cost"][0]) <= 1e-10 * c["cost"][0]      # the cost of the initial iterate is the Newton run's
    assert xx_star.shape == (6, 500) and uu_star.shape == (2, 500) and np.array_equal(uu_star[:, -1], uu_star[:, -2])
    # no convergence within max_iters: the last computed iterate is returned (optcon.py:137), whose cost the next
    # iteration would have printed: lower than the last printed one
    from aircraftoptimalcontrol_amd import batch as aoc
    bp = aoc.BatchProblem(pg["QQt"], pg["RRt"], pg["QQT"], pg["xx_ref"], pg["uu_ref"], float(pg["dt"]))
    us = uu_star.copy(); us[:, -1] = 0.0
    assert aoc.traj_cost(bp, xx_star[None], us[None])[0] < cost[-1]


def test_gradient_solve_stays_a_gradient_solve_when_it_compacts():
    """solve() re-packs the still-iterating trajectories into a smaller solver once fewer than half
    are active; that solver must be of the caller's class — a compacted GradientBatchSolver used to continue with
    Newton iterations.  A stopping threshold chosen so that ~60 % stop after two iterations makes the re-packing happen
    (B > compact_min); results with and without it must be identical."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 2200
    x0 = problems.perturbed_x0(pr, B, seed=12)
    probe = aoc.GradientBatchSolver(bp, B, aoc.make_params(stepsize_0=1e-1, armijo_maxiters=20))
    probe.set_initial_from_x0(x0)
    probe.iterate(0); probe.iterate(1)
    thr = float(np.percentile(probe.scalars()["descent"], 40))      # slopes are negative: 60 % lie above
    prm = aoc.make_params(max_iters=7, stepsize_0=1e-1, armijo_maxiters=20, term_cond=thr)
    res = []
    for compact in (True, False):
        s = aoc.GradientBatchSolver(bp, B, prm)
        s.set_initial_from_x0(x0)
        seen = []
        r = s.solve(compact=compact, callback=lambda kk, sc: seen.append(len(sc["cost"])))
        res.append((r, seen))
    (a, sa), (b, sb) = res
    assert min(sa) < B and min(sb) == B, (sa, sb)                    # the first run really re-packed
    assert a["converged"].sum() > 0.3 * B and len(set(sa)) >= 2     # some stopped early, the rest went on in a smaller batch
    for key in ("xx_star", "uu_star", "iters", "converged", "status"):
        assert np.array_equal(a[key], b[key], equal_nan=True), key
    for key in a["history"]:
        assert np.array_equal(a["history"][key], b["history"][key], equal_nan=True), key
    # and they are gradient iterations: every recorded slope is -sum |du|^2 < 0, after the re-packing too
    d = a["history"]["descent"]
    assert np.isfinite(d[:, -1]).any() and (d[np.isfinite(d)] < 0).all()
