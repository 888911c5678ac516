"""The drop-in modules (optcon, aircraft_simplified, lqr_tracking on the HIP library) used the way the
reference's scripts use the originals, against golden vectors from the reference.  Needs a GPU."""
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

DROPIN = os.path.join(ROOT, "aircraftoptimalcontrol_amd", "dropin")


@pytest.fixture(scope="module")
def mods():
    sys.path.insert(0, DROPIN)
    try:
        import aircraft_simplified
        import lqr_tracking
        import optcon
        yield aircraft_simplified, optcon, lqr_tracking
    finally:
        sys.path.remove(DROPIN)


def scaled_err(K, Kref):
    sc = np.abs(Kref).max(axis=-1, keepdims=True)
    return float(np.max(np.abs(K - Kref) / np.maximum(sc, 1e-300)))


def test_dynamics_step_and_cost_surface(mods):
    air, _, _ = mods
    g = load_golden("g1_step_dt0.002")
    dyn = air.Dynamics()
    dyn.dt = 2e-3
    assert (dyn.ns, dyn.ni, dyn.m, dyn.g, dyn.J) == (6, 2, 12, 9.81, 0.24)
    for i in range(4):
        xp, fx, fu, fxx, fuu, fux = dyn.step(g["x"][i], g["u"][i], g["lmbd"][i])
        assert xp.dtype == np.float32 and xp.shape == (6,) and np.array_equal(xp, g["xp"][i])
        assert fx.shape == (6, 6) and fu.shape == (2, 6) and fxx.shape == (6, 6) and fuu.shape == (2, 2) and fux.shape == (2, 6)
        assert rel_err(fx, g["fx"][i], 1e-12) < 1e-12 and rel_err(fu, g["fu"][i], 1e-12) < 1e-12
        assert np.allclose(fxx, g["fxx"][i], rtol=1e-11, atol=1e-13) and np.allclose(fux, g["fux"][i], rtol=1e-11, atol=1e-15)
        # without a costate: the uncontracted (6,6,6)/(2,2,6)/(2,6,6) tensors
        r = dyn.step(g["x"][i], g["u"][i])
        assert r[3].shape == (6, 6, 6) and r[4].shape == (2, 2, 6) and r[5].shape == (2, 6, 6)
        assert np.allclose(r[3], g["fxx3"][i], rtol=1e-12, atol=1e-16) and np.allclose(r[5], g["fux3"][i], rtol=1e-12, atol=1e-16)
    c = load_golden("g2_cost_step")
    cst = air.Cost(c["QQt"], c["RRt"], c["QQT"])
    for i in range(4):
        ll, lx, lu, lxx, lxu, lux, luu = cst.stagecost(c["x"][i], c["u"][i], c["xr"][i], c["ur"][i])
        assert ll.shape == (1, 1) and lx.shape == (6, 1) and lu.shape == (2, 1) and lxu.shape == (6, 2) and lux.shape == (2, 6)
        assert abs(ll.item() - c["ll"][i]) <= 1e-13 * abs(c["ll"][i]) and np.allclose(lx.squeeze(), c["lx"][i], rtol=1e-12)
        assert np.array_equal(lxx, c["QQt"]) and np.array_equal(luu, c["RRt"])
        llT, lTx, lTxx = cst.termcost(c["x"][i], c["xr"][i])
        assert llT.shape == (1, 1) and abs(llT.item() - c["llT"][i]) <= 1e-13 * abs(c["llT"][i])
        assert np.allclose(lTx.squeeze(), c["lTx"][i], rtol=1e-12)


def _lqr_inputs(prob_name, xx, uu, full):
    """A,B,Q,R,S,q,r along (xx,uu) in the reference's layout, built with the ORACLE's step/cost."""
    pg = load_golden(prob_name)
    op = orc.OracleProblem(pg["QQt"], pg["RRt"], pg["QQT"], pg["xx_ref"], pg["uu_ref"], float(pg["dt"]))
    mdl = orc.default_model(float(pg["dt"]))
    T = xx.shape[1]
    AA = np.zeros((6, 6, T)); BB = np.zeros((6, 2, T)); QQ = np.zeros((6, 6, T)); RR = np.zeros((2, 2, T))
    SS = np.zeros((2, 6, T)); qq = np.zeros((6, T)); rr = np.zeros((2, T)); lm = np.zeros((6, T))
    _, lT = orc.termcost(op, xx[:, -1], pg["xx_ref"][:, -1])
    lm[:, -1] = lT; qq[:, -1] = lT; QQ[:, :, -1] = pg["QQT"]
    for t in reversed(range(T - 1)):
        _, a, b = orc.stagecost(op, xx[:, t], uu[:, t], pg["xx_ref"][:, t], pg["uu_ref"][:, t])
        _, fx, fu, fxx, fuu, fux = orc.step(mdl, xx[:, t], uu[:, t], lm[:, t + 1])
        AA[:, :, t] = fx.T; BB[:, :, t] = fu.T
        QQ[:, :, t] = pg["QQt"] + (fxx if full else 0); RR[:, :, t] = pg["RRt"]; SS[:, :, t] = fux if full else 0
        qq[:, t] = a; rr[:, t] = b
        lm[:, t] = fx @ lm[:, t + 1] + a
    return AA, BB, QQ, RR, SS, qq, rr


@pytest.mark.parametrize("case,prob,tol", [("d_gn_init_T500", "problem_step_T500", 1e-8),
                                           ("c_full_near_T500", "problem_step_T500", 1e-8),
                                           ("b_full_init_T1000", "problem_step_T1000", 1e-5)])
def test_ltv_lqr_augmented_surface(mods, case, prob, tol):
    """optcon.ltv_LQR with affine terms: shapes, 'Augmented term!' print, KK/PP/dx/du vs the reference."""
    _, optcon, _ = mods
    g = load_golden("g3_lqr_" + case)
    T = g["xx"].shape[1]
    AA, BB, QQ, RR, SS, qq, rr = _lqr_inputs(prob, g["xx"], g["uu"], int(g["full_hessian"]))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        KK, PP, dx, du = optcon.ltv_LQR(AA, BB, QQ, RR, SS, QQ[:, :, -1], T, np.zeros(6), qq, rr, qq[:, -1])
    assert buf.getvalue() == "Augmented term!\n"
    assert KK.shape == (2, 7, T) and PP.shape == (7, 7, T) and dx.shape == (6, T) and du.shape == (2, T)
    assert scaled_err(KK, g["KK"]) < tol
    assert scaled_err(PP[:, :, ::50], g["PP_s"]) < tol
    assert rel_err(du, g["du"], 1e-3) < tol and scaled_err(dx, g["dx"]) < tol


def test_ltv_lqr_plain_and_tracking(mods):
    """Non-augmented ltv_LQR with 2-D weights broadcast over T (lqr_tracking.py:276) and
    lqr_tracking() itself on the reference's saved optimum (Data/xx_star.npy)."""
    air, optcon, trk = mods
    g = load_golden("g4_lqr_tracking")
    T = g["xx_opt"].shape[1]
    mdl = orc.default_model(float(g["dt"]))
    AA = np.zeros((6, 6, T)); BB = np.zeros((6, 2, T))
    for t in range(T):
        _, fx, fu, *_ = orc.step(mdl, g["xx_opt"][:, t], g["uu_opt"][:, t])
        AA[:, :, t] = fx.T; BB[:, :, t] = fu.T
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        KK, PP, lx, lu = optcon.ltv_LQR(AA, BB, g["QQt"], g["RRt"], np.zeros((2, 6, T)), g["QQT"], T, np.ones(6) * 0.1,
                                        None, None, None)
    assert buf.getvalue() == ""
    assert KK.shape == (2, 6, T) and PP.shape == (6, 6, T)
    assert scaled_err(KK, g["KK"]) < 1e-8 and scaled_err(PP[:, :, ::50], g["PP_s"]) < 1e-8
    assert scaled_err(lx, g["lin_xx"]) < 1e-8 and rel_err(lu, g["lin_uu"], 1e-3) < 1e-7
    assert trk.ltv_LQR is optcon.ltv_LQR
    xr, ur = trk.lqr_tracking(g["xx_opt"], g["uu_opt"], np.linspace(0, 1, T))
    assert xr.shape == (6, T) and ur.shape == (2, T)
    assert rel_err(ur, g["uu_reg"], 1e-3) < 1e-8
    assert np.array_equal(xr, g["xx_reg"])
    # dimension mismatch -> the reference's message and exit()
    with pytest.raises(SystemExit):
        with contextlib.redirect_stdout(io.StringIO()):
            optcon.ltv_LQR(AA, BB, np.eye(5), g["RRt"], np.zeros((2, 6, T)), g["QQT"], T, np.zeros(6))


def test_tracking_batch_vs_oracle(mods):
    """T1 batched: gains and closed-loop rollout for perturbed copies of the optimum vs the oracle."""
    from aircraftoptimalcontrol_amd import batch, problems
    g = load_golden("g4_lqr_tracking")
    T = g["xx_opt"].shape[1]
    rng = np.random.default_rng(4)
    B = 70
    XO = np.repeat(g["xx_opt"][None], B, 0); UO = np.repeat(g["uu_opt"][None], B, 0)
    UO = UO + rng.normal(0, 2.0, UO.shape)
    delta = rng.normal(0, 0.1, (B, 6))
    Q, R, QT = problems.tracking_weights()
    bp = batch.BatchProblem(Q, R, QT, np.zeros((6, T)), np.zeros((2, T)), float(g["dt"]))
    xr, ur, KK, st = batch.lqr_tracking_batch(bp, XO, UO, delta)
    mdl = orc.default_model(float(g["dt"]))
    for b in (0, 13, 69):
        xo, uo, Ko, ns = orc.lqr_tracking(mdl, Q, R, QT, XO[b], UO[b], delta[b])
        assert scaled_err(KK[b], Ko) < 1e-8
        assert rel_err(ur[b], uo, 1e-3) < 1e-8
        assert np.array_equal(xr[b], xo)
    assert not st.any()


def test_newton_method_like_main_script(mods):
    """The flow of main_newton_method.py:159-186 on the drop-in modules (tf = 1, dt = 2e-3): stdout lines,
    iteration count, returned optimum vs the reference's full solve (G8)."""
    air, optcon, _ = mods
    pg = load_golden("problem_step_T500")
    f = load_golden("g8_full_step_T500")
    dyn = air.Dynamics()
    dyn.dt = float(pg["dt"])
    cst = air.Cost(pg["QQt"], pg["RRt"], pg["QQT"])
    NM = optcon.NewtonMethod(dyn, cst, pg["xx_ref"], pg["uu_ref"], max_iters=200, stepsize_0=1, cc=0.5, beta=0.7,
                             armijo_maxiters=10, term_cond=1e-6, visu_armijo=False)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        xx_star, uu_star = NM.optimize(f["xx_init"], f["uu_init"], float(pg["tf"]), float(pg["dt"]))
    out = buf.getvalue()
    assert out.startswith("-*-*-*-*-*-\n") and "term = -1e-06" in out
    it = re.findall(r"Iter = (\d+)\t Descent = (\S+)\t Cost = (\S+)", out)
    steps = [float(s) for s in re.findall(r"Armijo stepsize = (\S+)", out)]
    n = int(f["n_done"])
    costs = np.array([float(c) for _, _, c in it]); desc = np.array([float(d) for _, d, _ in it])
    assert np.allclose(costs[:14], f["cost"][:14], rtol=1e-10, atol=0)
    assert np.allclose(desc[:14], f["descent"][:14], rtol=1e-8, atol=0)
    assert steps[:14] == f["stepsize"][:14].tolist()
    assert xx_star.shape == (6, 500) and uu_star.shape == (2, 500) and xx_star.dtype == np.float64
    if len(it) == n and np.array_equal(NM.stepsizes, f["stepsize"]):
        assert np.array_equal(xx_star, f["xx_star"]) and rel_err(uu_star, f["uu_star"], 1e-3) < 1e-8
        # exhausted searches print no 'Armijo stepsize' line (optcon.py:268-273)
        assert len(steps) == int(np.sum(f["ntrials"] < 10) + np.sum((f["ntrials"] == 10) & (f["stepsize"] > 0.7 ** 10 * 1.0000001)))
    else:
        assert abs(len(it) - n) <= 6 and abs(costs[-1] - f["cost"][-1]) <= 1e-6 * f["cost"][-1]
    assert np.array_equal(uu_star[:, -1], uu_star[:, -2])
    # a foreign dynamics object is refused, not silently mis-simulated
    class Other:
        ns, ni = 3, 1
    with pytest.raises(TypeError):
        optcon.NewtonMethod(Other(), cst, pg["xx_ref"], pg["uu_ref"]).optimize(f["xx_init"], f["uu_init"], 1.0, 2e-3)
