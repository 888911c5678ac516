"""The headless drivers (examples/) run as the reference's scripts are run — as processes that print and save
Data/*.npy — and are checked against what the reference's own drivers produced (BASELINE configs[0]):
main_newton_method.py:159-186 / acrobatic_newton.py:170-203 -> examples/run_newton.py,
lqr_tracking.py:321-342 -> examples/run_lqr_tracking.py."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, load_golden, rel_err

pytestmark = pytest.mark.gpu
ITER = re.compile(r"Iter = (\d+)\t Descent = (\S+)\t Cost = (\S+)")


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script)] + [str(a) for a in args],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def _check_saved(d, suffix, T):
    xs, us = np.load(os.path.join(d, "xx_star%s.npy" % suffix)), np.load(os.path.join(d, "uu_star%s.npy" % suffix))
    # what the reference's Data/*.npy are: (6,T) / (2,T) float64, C-order
    assert xs.shape == (6, T) and us.shape == (2, T) and xs.dtype == np.float64 and us.dtype == np.float64
    assert xs.flags.c_contiguous and us.flags.c_contiguous
    assert np.array_equal(us[:, -1], us[:, -2])           # optcon.py:505
    return xs, us


@pytest.mark.parametrize("problem,fixture,suffix", [("step", "g8_full_step_T1000", ""),
                                                    ("acrobatic", "g8_full_acro_T1000", "_acrobatic")])
def test_run_newton_from_the_references_initial_guess(tmp_path, problem, fixture, suffix):
    """The driver started from the reference's own initial guess (its get_initial_trajectory evaluates mostly in
    float32 under NumPy 2, which is not reproducible elsewhere — the guess is an INPUT here): stdout scalars of every
    iteration, iteration count, Armijo lines and the saved optimum against the reference's run."""
    f = load_golden(fixture)
    out = _run("run_newton.py", problem, "--out", tmp_path, "--init", os.path.join(GOLDEN, fixture + ".npz"))
    it = ITER.findall(out)
    n = int(f["n_done"])
    assert len(it) == n, "iterations: %d, reference %d" % (len(it), n)
    cost = np.array([float(c) for _, _, c in it]); desc = np.array([float(d) for _, d, _ in it])
    assert np.allclose(cost, f["cost"], rtol=1e-10, atol=0)
    assert np.allclose(desc, f["descent"], rtol=1e-6, atol=1e-9)
    steps = [float(v) for v in re.findall(r"Armijo stepsize = (\S+)", out)]
    exhausted = 0.7 ** 10 * (1 + 1e-12)
    assert steps == [v for v in f["stepsize"].tolist() if v > exhausted]   # exhausted searches print nothing (optcon.py:268-273)
    assert out.count("term = -1e-06") == n
    xs, us = _check_saved(tmp_path, suffix, 1000)
    assert np.array_equal(xs, f["xx_star"]) and rel_err(us, f["uu_star"], 1e-3) < 1e-8


@pytest.mark.parametrize("problem,fixture,suffix", [("step", "g8_full_step_T1000", ""),
                                                    ("acrobatic", "g8_full_acro_T1000", "_acrobatic")])
def test_run_newton_like_the_reference_scripts(tmp_path, problem, fixture, suffix):
    """`python main_newton_method.py` / `python acrobatic_newton.py` end to end, initial guess included (rolled out on
    the device in fp64 + float32 rounding: within 2e-4 of the reference's): same optimum — iteration count within a
    few, cost trace to 1e-3 while it is large, final cost to 1e-5, saved trajectory within the reference's own
    reproducibility (its committed Data/*.npy differ from a re-run by 5e-5 / 1.9e-4 in the states)."""
    f = load_golden(fixture)
    out = _run("run_newton.py", problem, "--out", tmp_path)
    it = ITER.findall(out)
    n = int(f["n_done"])
    assert abs(len(it) - n) <= 8, (len(it), n)
    cost = np.array([float(c) for _, _, c in it])
    assert np.allclose(cost[:10], f["cost"][:10], rtol=1e-3, atol=0)
    assert abs(cost[-1] - f["cost"][-1]) <= 1e-5 * f["cost"][-1]
    xs, us = _check_saved(tmp_path, suffix, 1000)
    assert np.abs(xs - f["xx_star"]).max() < 5e-3


def test_run_newton_animates_like_the_drivers_closing_call(tmp_path):
    """main_newton_method.py:226-227: Airfoil(20, xx_star, xx_ref, xlim=[0, 17], ylim=[-5, 5]).run_animation(name='step')
    at the end of the driver; every 100th of the 1000 frames here."""
    from PIL import Image
    figs = tmp_path / "Figures"
    out = _run("run_newton.py", "step", "--out", tmp_path, "--init", os.path.join(GOLDEN, "g8_full_step_T1000.npz"),
               "--animate", 100, "--figures", figs)
    assert "animation: %s" % (figs / "AircraftBehavior_step.gif") in out
    with Image.open(figs / "AircraftBehavior_step.gif") as im:
        assert im.n_frames == 10


def test_run_lqr_tracking_on_the_references_optimum(tmp_path):
    """`python lqr_tracking.py` (lqr_tracking.py:321-342) on the reference's committed Data/xx_star.npy, uu_star.npy."""
    g = load_golden("g4_lqr_tracking")
    np.save(tmp_path / "xx_star.npy", g["xx_opt"]); np.save(tmp_path / "uu_star.npy", g["uu_opt"])
    out = _run("run_lqr_tracking.py", "--data", tmp_path)
    assert "final tracking error" in out
    xl, ul = np.load(tmp_path / "xx_lqr.npy"), np.load(tmp_path / "uu_lqr.npy")
    assert np.array_equal(xl, g["xx_reg"]) and rel_err(ul, g["uu_reg"], 1e-3) < 1e-8


def test_run_batch(tmp_path):
    out = _run("run_batch.py", "--batch", 700, "--horizon", 500, "--max-iters", 40)
    m = re.search(r"trajectories (\d+)  converged (\d+)  iterations min/median/max (\d+)/(\d+)/(\d+)", out)
    assert m and int(m.group(1)) == 700 and int(m.group(2)) > 600 and 10 <= int(m.group(4)) <= 30
    c = re.search(r"cost first/last \(mean\): (\S+) -> (\S+)", out)
    assert float(c.group(2)) < float(c.group(1))


def test_visu_armijo_records_the_references_figure(tmp_path, monkeypatch):
    """NewtonMethod(..., visu_armijo=True): the reference's Armijo figure (optcon.py:280-325) — cost along the search
    direction at linspace(0, stepsize_0, armijo_maxiters), the tested steps — per iteration, against the oracle's
    get_update + cost on the same direction; the figure files and the two history figures (optcon.py:513-528)."""
    monkeypatch.setenv("MPLBACKEND", "Agg")
    monkeypatch.setenv("AOC_PLOT_DIR", str(tmp_path))
    sys.path.insert(0, os.path.join(ROOT, "aircraftoptimalcontrol_amd", "dropin"))
    try:
        import optcon
        import aircraft_simplified
    finally:
        sys.path.pop(0)
    from aircraftoptimalcontrol_amd import problems
    from oracle import oracle as orc
    pr = problems.step_maneuver(1.0, 2e-3)
    dyn = aircraft_simplified.Dynamics()
    dyn.dt = pr.dt
    cst = aircraft_simplified.Cost(pr.QQt, pr.RRt, pr.QQT)
    nm = optcon.NewtonMethod(dyn, cst, pr.xx_ref, pr.uu_ref, max_iters=4, stepsize_0=1, cc=0.5, beta=0.7,
                             armijo_maxiters=10, visu_armijo=True)
    xx0, uu0 = dyn.get_initial_trajectory(pr.xx_ref, pr.tt)
    nm.optimize(xx0, uu0, pr.tf, pr.dt)
    assert len(nm.armijo_curves) == 3
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    for kk, rec in enumerate(nm.armijo_curves):
        assert rec["steps"].shape == (10,) and rec["steps"][0] == 0 and rec["steps"][-1] == 1
        assert rec["costs"][0] == rec["JP"] == nm.JJ[kk]           # step 0: the cost of the iterate itself (Q10)
        assert rec["costs_armijo"][-1] <= rec["JP"] + 0.5 * rec["stepsizes"][-1] * rec["descent"]   # the accepted trial
        assert os.path.exists(os.path.join(tmp_path, "armijo_iter%d.png" % kk))
    # the curve of iteration 0 against the oracle: same direction, the reference's rollout and cost
    r = orc.newton_iterate(op, orc.params(), 0, xx0[:, :pr.T], uu0[:, :pr.T], xx0[:, 0], want_internals=True)
    rec = nm.armijo_curves[0]
    for ii in (3, 9):
        xo, uo = orc.get_update(op, float(rec["steps"][ii]), uu0[:, :pr.T], r["du"], xx0[:, 0])
        assert abs(orc.traj_cost(op, xo, uo) - rec["costs"][ii]) <= 1e-9 * abs(rec["costs"][ii])
    f1, f2 = nm.plot_histories()
    assert len(f1.axes[0].lines) == 1 and len(f2.axes[0].lines) == 1
