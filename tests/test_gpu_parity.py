"""Parity of the HIP path (through the C-ABI) against the CPU oracle and the golden vectors captured
from the reference.  Needs a real MI355X: run with  -m gpu.

Tolerances: unit quantities 1e-12 rel; gains/direction/iterates 1e-8 rel; rollouts with
the reference's float32 state rounding bit-exact.  The float32 rounding makes whole-trajectory
comparison discontinuous: a last-bit difference in u can flip one rounding and move every later
state by an fp32 ulp.  Chains are therefore checked both free-running and "teacher-forced" (every
iteration restarted from the golden iterate), and flips — if any — are counted, not hidden."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aoc():
    from aircraftoptimalcontrol_amd import batch
    return batch


def scaled_err(K, Kref, axis=-1):
    sc = np.abs(Kref).max(axis=axis, keepdims=True)
    return float(np.max(np.abs(K - Kref) / np.maximum(sc, 1e-300)))


def _problem(aoc, name):
    g = load_golden(name)
    bp = aoc.BatchProblem(g["QQt"], g["RRt"], g["QQT"], g["xx_ref"], g["uu_ref"], float(g["dt"]))
    op = orc.OracleProblem(g["QQt"], g["RRt"], g["QQT"], g["xx_ref"], g["uu_ref"], float(g["dt"]))
    return g, bp, op


# ----------------------------------------------------------------------------------------------
# unit level: D1 (Dynamics.step), C1/C2 (Cost)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", [1e-3, 2e-3])
def test_step_batch_vs_golden(aoc, dt):
    g = load_golden("g1_step_dt%g" % dt)
    xp, fx, fu, fxx, fuu, fux = aoc.step_batch(aoc.default_model(dt), g["x"], g["u"], g["lmbd"])
    assert np.array_equal(xp.astype(np.float32), g["xp"])  # fp32-rounded next state: bit-exact
    assert rel_err(fx, g["fx"], 1e-12) < 1e-12
    assert rel_err(fu, g["fu"], 1e-12) < 1e-12
    assert scaled_err(fxx.reshape(-1, 36), g["fxx"].reshape(-1, 36)) < 1e-12
    assert scaled_err(fux.reshape(-1, 12), g["fux"].reshape(-1, 12)) < 1e-12
    assert not fuu.any()


def test_step_batch_vs_oracle_random(aoc):
    rng = np.random.default_rng(7)
    n = 4096
    X = np.stack([rng.uniform(-5, 20, n), rng.uniform(-3, 5, n), rng.uniform(9, 23, n),
                  rng.uniform(-1.5, 1.5, n), rng.uniform(-25, 25, n), rng.uniform(-1.5, 1.5, n)], 1)
    U = np.stack([rng.uniform(0, 900, n), rng.uniform(-100, 300, n)], 1)
    mdl = orc.default_model(2e-3)
    xp = aoc.step_batch(aoc.default_model(2e-3), X, U)[0]
    ref = np.stack([orc.step(mdl, X[i], U[i])[0] for i in range(n)])
    assert np.array_equal(xp, ref)


@pytest.mark.parametrize("tag", ["step", "acro"])
def test_cost_batch_vs_golden(aoc, tag):
    g = load_golden("g2_cost_" + tag)
    bp = aoc.BatchProblem(g["QQt"], g["RRt"], g["QQT"], np.zeros((6, 4)), np.zeros((2, 4)), 1e-3)
    ll, lx, lu, llT, lTx = aoc.cost_batch(bp, g["x"], g["u"], g["xr"], g["ur"])
    assert np.allclose(ll, g["ll"], rtol=1e-13, atol=0) and np.allclose(llT, g["llT"], rtol=1e-13, atol=0)
    assert rel_err(lx, g["lx"], 1e-12) < 1e-13 and rel_err(lu, g["lu"], 1e-12) < 1e-13
    assert rel_err(lTx, g["lTx"], 1e-12) < 1e-13


def test_cost_batch_dense_weights(aoc):
    """Cost accepts general dense Q/R (aircraft_simplified.py:61); exercised vs the oracle."""
    rng = np.random.default_rng(3)
    A = rng.normal(size=(6, 6)); Q = A @ A.T
    Bm = rng.normal(size=(2, 2)); R = Bm @ Bm.T
    A2 = rng.normal(size=(6, 6)); QT = A2 @ A2.T
    bp = aoc.BatchProblem(Q, R, QT, np.zeros((6, 4)), np.zeros((2, 4)), 1e-3)
    op = orc.OracleProblem(Q, R, QT, np.zeros((6, 4)), np.zeros((2, 4)), 1e-3)
    n = 100
    x, xr = rng.normal(size=(n, 6)), rng.normal(size=(n, 6))
    u, ur = rng.normal(size=(n, 2)), rng.normal(size=(n, 2))
    ll, lx, lu, llT, lTx = aoc.cost_batch(bp, x, u, xr, ur)
    for i in range(n):
        l0, lx0, lu0 = orc.stagecost(op, x[i], u[i], xr[i], ur[i])
        lT0, lTx0 = orc.termcost(op, x[i], xr[i])
        assert abs(ll[i] - l0) <= 1e-13 * abs(l0) and abs(llT[i] - lT0) <= 1e-13 * abs(lT0)
        assert np.allclose(lx[i], lx0, rtol=1e-12, atol=1e-14) and np.allclose(lTx[i], lTx0, rtol=1e-12, atol=1e-14)


def test_pack_unpack_roundtrip(aoc):
    import torch
    rng = np.random.default_rng(0)
    for B in (1, 63, 64, 65, 200):
        a = rng.normal(size=(B, 6, 37))
        t = aoc.pack(a)
        assert tuple(t.shape) == (aoc.ntiles(B), 37, 6, 64)
        assert np.array_equal(aoc.unpack(t, B).cpu().numpy(), a)
        # padded lanes replicate trajectory B-1
        if B % 64:
            assert torch.equal(t[-1, :, :, B % 64:], t[-1, :, :, (B % 64) - 1:(B % 64)].expand(-1, -1, 64 - B % 64))


# ----------------------------------------------------------------------------------------------
# pass level: A1/A2 rollout + cost, N1-C/L1 backward + forward
# ----------------------------------------------------------------------------------------------
def test_rollout_cost_vs_oracle(aoc):
    g, bp, op = _problem(aoc, "problem_step_T500")
    c = load_golden("g6_chain_step_T500")
    rng = np.random.default_rng(11)
    B = 96
    uu = np.repeat(c["uu_it3"][None], B, 0) + rng.normal(0, 1.0, (B, 2, 500))
    du = rng.normal(0, 1.0, (B, 2, 500))
    x0 = c["xx_init"][:, 0][None] + rng.normal(0, 0.1, (B, 6))
    alpha = 0.7 ** rng.integers(0, 10, B)
    xx, un, J, st = aoc.rollout_cost(bp, x0, uu, du, alpha)
    flips = 0
    for b in range(B):
        xo, uo = orc.get_update(op, alpha[b], uu[b], du[b], x0[b])
        Jo = orc.traj_cost(op, xo, uo)
        assert np.array_equal(un[b], uo)
        flips += int(not np.array_equal(xx[b], xo))
        if np.array_equal(xx[b], xo):
            assert J[b] == Jo
    assert flips == 0, "fp32-rounded rollouts must be bit-identical (%d of %d differ)" % (flips, B)
    assert not st.any()
    # cost-only variant and the stored-trajectory cost agree with the rollout's own cost
    _, _, J2, _ = aoc.rollout_cost(bp, x0, uu, du, alpha, write=False)
    assert np.array_equal(J, J2)
    assert np.array_equal(aoc.traj_cost(bp, xx, un), J)
    # float32 state storage on the device returns the same trajectory
    xf, uf, Jf, _ = aoc.rollout_cost(bp, x0, uu, du, alpha, f32=True)
    assert np.array_equal(xf, xx) and np.array_equal(uf, un) and np.array_equal(Jf, J)


LQR_CASES = [("a_gn_init_T1000", "problem_step_T1000", 1e-8), ("b_full_init_T1000", "problem_step_T1000", 1e-5),
             ("c_full_near_T500", "problem_step_T500", 1e-8), ("d_gn_init_T500", "problem_step_T500", 1e-8)]


@pytest.mark.parametrize("f32", [False, True])
@pytest.mark.parametrize("case,prob,tol", LQR_CASES)
def test_backward_forward_vs_golden(aoc, case, prob, tol, f32):
    """Fused backward pass (costate + quadratisation + Riccati + gains) and LQR rollout against the
    reference's ltv_LQR outputs, incl. the regularisation branch (case b: 18 stages); with the state
    trajectory handed over as fp64 and as float32 (identical values: the iterates are float32-valued)."""
    g = load_golden("g3_lqr_" + case)
    _, bp, op = _problem(aoc, prob)
    fh = int(g["full_hessian"])
    r = aoc.backward_forward(bp, g["xx"][None], g["uu"][None], fh, f32=f32)
    KK = r["KK"][0]
    assert scaled_err(KK[:, :, :-1], g["KK"][:, :, :-1]) < tol
    assert rel_err(r["du"][0], g["du"], 1e-3) < tol
    assert abs(r["descent"][0] - float(g["descent"])) <= tol * abs(float(g["descent"]))
    assert rel_err(r["lmbd0"][0], g["lmbd"][:, 0], 1e-9) < 1e-9
    from aircraftoptimalcontrol_amd import _lib
    assert bool(r["status"][0] & _lib.ST_REGULARISED) == (int(g["n_regularised"]) > 0)
    # cost of the first Armijo trial = cost of the oracle's get_update(stepsize_0) on the SAME du
    xo, uo = orc.get_update(op, 1.0, g["uu"], r["du"][0], g["xx"][:, 0])
    assert r["J_trial0"][0] == orc.traj_cost(op, xo, uo)
    if f32:  # both storage types give bit-identical results
        r64 = aoc.backward_forward(bp, g["xx"][None], g["uu"][None], fh, f32=False)
        for key in ("KK", "du", "descent", "J_trial0", "lmbd0"):
            assert np.array_equal(r[key], r64[key]), key


# ----------------------------------------------------------------------------------------------
# iteration level: N1 (optimize), A1 (armijo), chains
# ----------------------------------------------------------------------------------------------
CHAINS = [("g6_chain_step_T500", "problem_step_T500"), ("g6_chain_step_T1000", "problem_step_T1000"),
          ("g6_chain_acro_T1000", "problem_acro_T1000")]


def _flip_aware_compare(xx, uu, gx, gu, tag):
    """-> True if clean (xx bit-identical, uu within 1e-8); otherwise asserts the deviation is of
    fp32-rounding-flip size (states within a few fp32 ulps) and returns False."""
    if np.array_equal(xx, gx):
        assert rel_err(uu, gu, 1e-3) < 1e-8, tag
        return True
    assert rel_err(xx, gx, 1e-2) < 5e-6, "%s: state deviation larger than fp32 rounding flips" % tag
    return False


@pytest.mark.parametrize("chain,prob", CHAINS)
def test_chain_teacher_forced(aoc, chain, prob):
    """Every stored golden iterate k -> k+1 with one HIP Newton iteration started from the GOLDEN
    iterate k: Armijo step, trial count, cost, descent and the new iterate."""
    g, bp, op = _problem(aoc, prob)
    c = load_golden(chain)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    keys = sorted(int(k[5:]) for k in c if k.startswith("xx_it"))
    pairs = [(0, 1)] + [(k, k + 1) for k in keys if k + 1 in keys]
    flips = 0
    for k0, k1 in pairs:
        xx = c["xx_init"] if k0 == 0 else c["xx_it%d" % k0]
        uu = c["uu_init"] if k0 == 0 else c["uu_it%d" % k0]
        s = aoc.NewtonBatchSolver(bp, 1, prm)
        xi = xx[None].copy(); xi[0, :, 0] = c["xx_init"][:, 0]
        s.set_initial(xi, uu[None])
        s.iterate(k0)
        sc = s.scalars()
        assert abs(sc["cost"][0] - c["cost"][k0]) <= 1e-10 * abs(c["cost"][k0]), k0
        assert abs(sc["descent"][0] - c["descent"][k0]) <= 1e-8 * abs(c["descent"][k0]), k0
        assert sc["stepsize"][0] == c["stepsize"][k0], k0
        assert sc["ntrials"][0] == c["ntrials"][k0], k0
        xn, un = s.current()
        flips += int(not _flip_aware_compare(xn[0], un[0], c["xx_it%d" % k1], c["uu_it%d" % k1], "iter %d" % k0))
    assert flips == 0, "%d of %d iterations showed an fp32 rounding flip" % (flips, len(pairs))


@pytest.mark.parametrize("chain,prob", CHAINS)
def test_chain_free_running(aoc, chain, prob):
    g, bp, op = _problem(aoc, prob)
    c = load_golden(chain)
    n = int(c["n_done"])
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, 1, prm)
    s.set_initial(c["xx_init"][None], c["uu_init"][None])
    clean = True
    for kk in range(n):
        s.iterate(kk)
        sc = s.scalars()
        if clean:
            assert abs(sc["cost"][0] - c["cost"][kk]) <= 1e-10 * abs(c["cost"][kk]), kk
            assert abs(sc["descent"][0] - c["descent"][kk]) <= 1e-8 * abs(c["descent"][kk]), kk
            assert sc["stepsize"][0] == c["stepsize"][kk], kk
        else:  # after a flip only the cost trace is comparable (rel 1e-6)
            assert abs(sc["cost"][0] - c["cost"][kk]) <= 1e-6 * abs(c["cost"][kk]), kk
        key = "xx_it%d" % (kk + 1)
        if key in c and clean:
            xn, un = s.current()
            clean = _flip_aware_compare(xn[0], un[0], c[key], c["uu_it%d" % (kk + 1)], "iter %d" % kk)
    assert clean, "free-running chain left the bit-exact path (fp32 flip) before iteration %d" % n


def test_minibatch_g9_and_lane_independence(aoc):
    """Config-2 generation rule mini-batch (perturbed x0): 4 iterations, all trajectories in ONE
    launch, plus: results do not depend on the lane/tile a trajectory sits in."""
    g, bp, op = _problem(aoc, "problem_step_T500")
    m = load_golden("g9_minibatch_step_T500")
    n = int(m["n_iters"])
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    Bm = m["xx_init"].shape[0]
    reps = 9  # 72 trajectories -> two tiles, second one ragged
    perm = np.random.default_rng(5).permutation(Bm * reps)
    XI = np.tile(m["xx_init"], (reps, 1, 1))[perm]
    UI = np.tile(m["uu_init"], (reps, 1, 1))[perm]
    s = aoc.NewtonBatchSolver(bp, Bm * reps, prm)
    s.set_initial(XI, UI)
    hist = s.run_fixed(n)
    xx, uu = s.current()
    src = np.tile(np.arange(Bm), reps)[perm]
    for j in range(Bm * reps):
        b = src[j]
        for it in range(n):
            assert hist[it]["stepsize"][j] == m["stepsize"][b, it]
            assert abs(hist[it]["cost"][j] - m["cost"][b, it]) <= 1e-10 * abs(m["cost"][b, it])
            assert abs(hist[it]["descent"][j] - m["descent"][b, it]) <= 1e-8 * abs(m["descent"][b, it])
        assert np.array_equal(xx[j, :, 1:].astype(np.float32), m["xx_out"][b, :, 1:])
        assert rel_err(uu[j], m["uu_out"][b], 1e-3) < 1e-8
    # identical inputs in different lanes give bit-identical outputs
    for b in range(Bm):
        idx = np.nonzero(src == b)[0]
        for j in idx[1:]:
            assert np.array_equal(uu[j], uu[idx[0]]) and np.array_equal(xx[j], xx[idx[0]])


G8 = [("g8_full_step_T500", "problem_step_T500"), ("g8_full_step_T1000", "problem_step_T1000"),
      ("g8_full_acro_T1000", "problem_acro_T1000")]
# Iterations over which the HIP path's Armijo history (step, trial count) must equal the reference's, per fixture.
# The whole history: the build reproduces the reference's solve verdict for verdict, so the returned iterate is
# compared with the reference's xx_star / uu_star directly.  (If a kernel change ever moves a late, noise-level
# verdict this number has to be lowered explicitly — the test then still pins the returned iterate, see below.)
G8_IDENTICAL_HISTORY = {"g8_full_step_T500": 22, "g8_full_step_T1000": 23, "g8_full_acro_T1000": 37}


@pytest.mark.parametrize("device_loop", [False, True])
@pytest.mark.parametrize("name,prob", G8)
def test_full_solve_termination_and_return_index(aoc, name, prob, device_loop):
    """G8 / BASELINE configs[0]: the reference's drivers solved to their own termination — main_newton_method.py at
    T = 1000 (23 iterations, returned iterate 21), acrobatic_newton.py (37), and the T = 500 variant (22) — through
    the host loop and through aoc_newton_solve: iteration count, returned iterate two behind the newest (Q7),
    uu[:,-1] = uu[:,-2] (Q8), Armijo exhaustion (Q5) at the recorded iterations."""
    g, bp, op = _problem(aoc, prob)
    f = load_golden(name)
    prm = aoc.make_params(max_iters=200, stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, 3, prm)
    s.set_initial(np.repeat(f["xx_init"][None], 3, 0), np.repeat(f["uu_init"][None], 3, 0))
    r = s.solve_on_device(sync_every=2) if device_loop else s.solve()
    n = int(f["n_done"])
    h = r["history"]
    m = min(n, int(r["iters"][0]))
    same = (h["stepsize"][0, :m] == f["stepsize"][:m]) & (h["ntrials"][0, :m] == f["ntrials"][:m])
    n_same = m if same.all() else int(np.argmin(same))
    assert n_same >= G8_IDENTICAL_HISTORY[name], "Armijo history leaves the reference's at iteration %d" % n_same
    assert np.allclose(h["cost"][0, :n_same], f["cost"][:n_same], rtol=1e-10, atol=0)
    assert np.allclose(h["descent"][0, :n_same], f["descent"][:n_same], rtol=1e-6, atol=1e-9)
    k_ret = int(r["iters"][0]) - 2                       # history index the solve returned (Q7)
    if n_same == n:                                      # the whole solve equals the reference's
        assert r["iters"][0] == n
        assert np.array_equal(r["xx_star"][0], f["xx_star"])
        assert rel_err(r["uu_star"][0], f["uu_star"], 1e-3) < 1e-8
    else:  # late iterations wander in fp32 noise: count +-, final cost 1e-6, and the returned iterate
        #    against the reference's iterate of the SAME index where the fixture holds it, within fp32-flip size
        assert abs(int(r["iters"][0]) - n) <= 6
        assert abs(h["cost"][0, int(r["iters"][0]) - 1] - f["cost"][-1]) <= 1e-6 * f["cost"][-1]
        if "xx_it%d" % k_ret in f:
            assert rel_err(r["xx_star"][0], f["xx_it%d" % k_ret], 1e-2) < 5e-6
    assert np.array_equal(r["uu_star"][0][:, -1], r["uu_star"][0][:, -2])
    assert r["converged"].all()
    from aircraftoptimalcontrol_amd import _lib
    assert (r["status"] & _lib.ST_CONVERGED).all()
    assert (r["status"][0] & _lib.ST_ARMIJO_EXH) != 0  # the reference exhausts the line search several times here
    # the three identical trajectories agree bit for bit
    assert np.array_equal(r["xx_star"][0], r["xx_star"][1]) and np.array_equal(r["uu_star"][0], r["uu_star"][2])


def test_initial_trajectory_on_device(aoc):
    """Row 8f-1: the P-controller initial guess rolled out on the device equals the oracle's
    (fp64 + float32 state rounding) bit for bit, and the reference's own (mostly-float32) to ~1e-4."""
    g, bp, op = _problem(aoc, "problem_step_T500")
    m = load_golden("g9_minibatch_step_T500")
    B = m["x0"].shape[0]
    s = aoc.NewtonBatchSolver(bp, B, aoc.make_params(stepsize_0=1.0, armijo_maxiters=10))
    s.set_initial_from_x0(m["x0"])
    xx, uu = s.current()
    mdl = orc.default_model(2e-3)
    for b in range(B):
        xr = g["xx_ref"].copy(); xr[:, 0] = m["x0"][b]
        xo, uo = orc.initial_trajectory(mdl, xr)
        assert np.array_equal(xx[b], xo) and np.array_equal(uu[b], uo)
        assert rel_err(xx[b], m["xx_init"][b], 1e-2) < 2e-4


LS_MODES = {"rounds": dict(ls_worklist=0), "rounds-wcap8": dict(ls_worklist=0, ls_wcap=8, ls_kgrow=1),
            "worklist-cpl4": dict(ls_worklist=1, ls_cpl=4), "worklist-cpl2": dict(ls_worklist=1, ls_cpl=2),
            "worklist-cpl1-depth1": dict(ls_worklist=1, ls_cpl=1, ls_depth_min=1)}


@pytest.mark.parametrize("mode", sorted(LS_MODES))
def test_linesearch_rounds_vs_oracle(aoc, mode, tuned):
    """A1: every scheduling of the back-tracking — rounds over a compacted list with several candidate steps of
    one trajectory evaluated at once when few trajectories still search (also with a round capacity of 8 wavefronts:
    four rounds); "worklist": (trajectory, up to cpl candidates) items,
    depth predicted from the previous iteration's trial count — accepts exactly the step and reports exactly the
    trial count of the reference's sequential loop; checked against the oracle's armijo_stepsize on the GPU's own
    iterates, over four iterations."""
    from aircraftoptimalcontrol_amd import problems
    tuned(nspec=2, **LS_MODES[mode])   # a batch this small would otherwise try every step in the forward pass
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 1000
    x0 = problems.random_x0(B, seed=99)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    oprm = orc.params()
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    sample = list(range(0, B, 23))
    seen = set()
    for kk in range(4):
        xi, ui = s.current()
        s.iterate_timed(kk) if kk % 2 else s.iterate(kk)   # both launch routes
        sc = s.scalars()
        xn, un = s.current()
        for b in sample:
            r = orc.newton_iterate(op, oprm, kk, xi[b], ui[b], xi[b][:, 0])
            assert r["stepsize"] == sc["stepsize"][b] and r["ntrials"] == sc["ntrials"][b], (kk, b)
            assert abs(r["J"] - sc["cost"][b]) <= 1e-12 * abs(r["J"])
            assert rel_err(un[b], r["uu"], 1e-3) < 1e-8
            seen.add(int(r["ntrials"]))
    assert len(seen) >= 4, "the sample should exercise several trial counts: %s" % sorted(seen)


@pytest.mark.parametrize("horizon_cut", [False, True])
def test_shard_invariance_and_large_batch(aoc, tuned, horizon_cut):
    """8(e): results do not depend on the batch a trajectory is solved in.  Trajectories 0..639 of a
    66 000-trajectory launch (line-search rounds over >1000 wavefronts, one step per round) equal the
    same trajectories solved in a 640-trajectory launch (few wavefronts, many steps per round) bit
    for bit; x0 is keyed by the global trajectory index.
    horizon_cut (the default for batches of at most 64 tiles, aoc_tuning.bw_hcut): the Gauss-Newton backward pass of
    the small launch evaluates the Riccati recursion in horizon segments — another order of the same arithmetic — so
    there the comparison is: costs and inputs equal to 1e-12 (measured 4e-14), float32 states identical except where an
    input that moved in its 14th digit tips a rounding to float32 (measured: 0 or 1 state of 1.9 million, one float32
    ulp, depending on where the cuts fall — tools/probes/cut_state_flips.py); with the cut switched off everything is
    bit for bit again."""
    from aircraftoptimalcontrol_amd import problems
    if not horizon_cut:
        tuned(bw_hcut=0, track_hcut=0)
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    out = []
    for B, first in ((66000, 0), (640, 0), (640, 65000)):
        s = aoc.NewtonBatchSolver(bp, B, prm)
        s.set_initial_from_x0(problems.random_x0(B, seed=20260403, first=first))
        h = s.run_fixed(3)
        out.append((h, s.current()))
    (hb, (xb, ub)), (hs, (xs, us)), (ht, (xt, ut)) = out
    for (h2, x2, u2, lo) in ((hs, xs, us, 0), (ht, xt, ut, 65000)):
        if horizon_cut:
            a, b = xb[lo:lo + 640], x2
            assert np.array_equal(a[:, :, 0], b[:, :, 0])                              # x0 itself (fp64)
            a, b = a[:, :, 1:].astype(np.float32), b[:, :, 1:].astype(np.float32)      # the stored states are float32 values
            ulps = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
            assert ulps.max() <= 1 and np.count_nonzero(ulps) <= 1e-5 * ulps.size, (ulps.max(), np.count_nonzero(ulps))
            ref = ub[lo:lo + 640]
            assert (np.abs(ref - u2).max(2) / np.maximum(np.abs(ref).max(2), 1e-3)).max() < 1e-12
            assert not np.array_equal(ref, u2), "the small launch is expected to take the horizon cut"
        else:
            assert np.array_equal(xb[lo:lo + 640], x2)
            assert np.array_equal(ub[lo:lo + 640], u2)
        for a, b in zip(hb, h2):
            for key in ("stepsize", "ntrials", "cost", "cost_new", "descent"):
                if horizon_cut and key in ("cost", "cost_new", "descent"):
                    assert np.allclose(a[key][lo:lo + 640], b[key], rtol=1e-11, atol=0), key
                else:
                    assert np.array_equal(a[key][lo:lo + 640], b[key]), key
    # size-independent properties at scale: Armijo holds for every accepted step, costs decrease
    last = hb[-1]
    acc = last["ntrials"] < 10
    assert np.all(last["cost_new"][acc] <= last["cost"][acc] + 0.5 * last["stepsize"][acc] * last["descent"][acc])
    assert np.all(last["descent"] < 0) and not last["status"].any()


def test_initial_iterate_with_arbitrary_fp64_samples(aoc):
    """The reference linearises iteration 0 about xx_init as given (optcon.py:395, :436-437), which need
    not be float32-valued.  Such an iterate is kept in fp64 for the first iteration; checked against
    the oracle on an initial state trajectory that is the (smooth, fp64) reference curve itself, and
    returned verbatim when a trajectory stops after one iteration."""
    g, bp, op = _problem(aoc, "problem_step_T500")
    c = load_golden("g6_chain_step_T500")
    xx0 = g["xx_ref"].copy() + 1e-3 * np.sin(np.arange(500))[None, :]   # not float32-representable
    xx0[:, 0] = c["xx_init"][:, 0]
    uu0 = c["uu_init"].copy()
    assert not np.array_equal(xx0[:, 1:].astype(np.float32).astype(np.float64), xx0[:, 1:])
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, 2, prm)
    s.set_initial(np.stack([xx0, c["xx_init"]]), np.stack([uu0, c["uu_init"]]))
    assert s.cur_is64
    xi, ui = s.current()
    assert np.array_equal(xi[0], xx0) and np.array_equal(xi[1], c["xx_init"])
    s.iterate(0)
    sc = s.scalars()
    xn, un = s.current()
    oprm = orc.params()
    for b, (xa, ua) in enumerate(((xx0, uu0), (c["xx_init"], c["uu_init"]))):
        r = orc.newton_iterate(op, oprm, 0, xa, ua, xa[:, 0])
        assert r["stepsize"] == sc["stepsize"][b] and r["ntrials"] == sc["ntrials"][b]
        assert abs(r["J"] - sc["cost"][b]) <= 1e-12 * abs(r["J"])
        assert abs(r["descent"] - sc["descent"][b]) <= 1e-8 * abs(r["descent"])
        assert np.array_equal(xn[b], r["xx"]) and rel_err(un[b], r["uu"], 1e-3) < 1e-8
    # stop after iteration 1 (term_cond huge): the reference returns history index 0 = xx_init itself
    prm2 = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10, term_cond=-1e30)
    s2 = aoc.NewtonBatchSolver(bp, 2, prm2)
    s2.set_initial(np.stack([xx0, c["xx_init"]]), np.stack([uu0, c["uu_init"]]))
    r2 = s2.solve()
    # descent >= -1e30 at kk = 0 already: index -1 = the all-zero slot (Q7)
    assert r2["iters"].tolist() == [1, 1] and not r2["xx_star"].any() and not r2["uu_star"].any()


def test_solve_returns_initial_iterate_when_stopping_at_kk1(aoc):
    """Q7 at its edge: a trajectory whose descent passes the threshold at kk = 1 returns history index 0,
    i.e. the caller's initial iterate itself (with uu[:,-1] = uu[:,-2], Q8) — also when that iterate is
    not float32-valued and lives in the fp64 side buffer."""
    g, bp, op = _problem(aoc, "problem_step_T500")
    c = load_golden("g6_chain_step_T500")
    d0, d1 = c["descent"][0], c["descent"][1]
    assert d0 < d1 < 0
    thr = 0.5 * (d0 + d1)                      # descent[0] < thr <= descent[1]
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10, term_cond=thr)
    s = aoc.NewtonBatchSolver(bp, 1, prm)
    s.set_initial(c["xx_init"][None], c["uu_init"][None])
    r = s.solve()
    assert r["iters"][0] == 2 and r["converged"][0]
    assert np.array_equal(r["xx_star"][0], c["xx_init"])
    exp_u = c["uu_init"].copy(); exp_u[:, -1] = exp_u[:, -2]
    assert np.array_equal(r["uu_star"][0], exp_u)
    # same with an initial iterate that is not float32-representable
    xx0 = c["xx_init"].copy(); xx0[:, 1:] += 1e-9
    s.set_initial(xx0[None], c["uu_init"][None])
    assert s.cur_is64
    r = s.solve()
    if r["iters"][0] == 2:
        assert np.array_equal(r["xx_star"][0], xx0)


def test_solve_with_repacking_equals_solve_without(aoc):
    """solve() re-packs the trajectories that are still iterating into a smaller batch once more than
    half have converged; per-trajectory results, iteration counts and histories must not change."""
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 2500
    x0 = problems.perturbed_x0(pr, B, seed=21)
    prm = aoc.make_params(max_iters=60, stepsize_0=1.0, armijo_maxiters=10)
    res = []
    for compact in (False, True):
        s = aoc.NewtonBatchSolver(bp, B, prm)
        s.set_initial_from_x0(x0)
        res.append(s.solve(compact=compact, compact_min=256))
    a, b = res
    assert np.array_equal(a["iters"], b["iters"]) and np.array_equal(a["converged"], b["converged"])
    assert len(set(a["iters"].tolist())) > 5          # a spread of iteration counts, so re-packing happened
    assert np.array_equal(a["xx_star"], b["xx_star"]) and np.array_equal(a["uu_star"], b["uu_star"])
    assert np.array_equal(a["status"], b["status"])
    for key in ("cost", "descent", "stepsize"):
        hb = b["history"][key]
        ha = a["history"][key][:, :hb.shape[1]]
        live = ~np.isnan(hb)
        assert np.array_equal(ha[live], hb[live]), key
        # a trajectory's history ends exactly where it stopped iterating
        assert np.array_equal(live.sum(1), b["iters"])


def _same_solve(a, b):
    assert a["last_kk"] == b["last_kk"]
    for key in ("iters", "converged", "status"):
        assert np.array_equal(a[key], b[key]), key
    for key in ("xx_star", "uu_star"):      # a diverged trajectory is NaN in both
        assert np.array_equal(a[key], b[key], equal_nan=True), key
    for key in ("cost", "descent", "stepsize", "ntrials"):
        assert np.array_equal(a["history"][key], b["history"][key], equal_nan=(key != "ntrials")), key


@pytest.mark.parametrize("worklist", [0, 1])
@pytest.mark.parametrize("sync_every", [0, 3])
def test_device_solve_equals_host_loop(aoc, sync_every, worklist, tuned):
    """aoc_newton_solve (loop, stopping rule Q6, return index Q7, Q8 on the device) against solve(compact=False),
    the host loop over aoc_newton_iterate: returned iterates, iteration counts, status flags and histories
    bit for bit.  max_iters is chosen so that part of the batch stops by the descent test and the rest runs out
    of iterations; sync_every = 0 never looks at the device between iterations."""
    from aircraftoptimalcontrol_amd import problems
    if worklist:   # the large-batch arrangement: one-wavefront kernels, two speculated trials, work-list search (frozen lanes!)
        tuned(ls_worklist=1, nspec=2, split_tiles=0, split_bw_tiles=0)
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 700
    x0 = problems.perturbed_x0(pr, B, seed=5)
    prm = aoc.make_params(max_iters=24, stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    host = s.solve(compact=False)
    s.set_initial_from_x0(x0)
    dev = s.solve_on_device(sync_every=sync_every)
    assert 0 < host["converged"].sum() < B, "the case should mix stopped and unfinished trajectories"
    if sync_every == 0:
        assert dev["last_kk"] == prm.max_iters - 2
        dev["last_kk"] = host["last_kk"]
        for key in dev["history"]:
            dev["history"][key] = dev["history"][key][:, :host["history"][key].shape[1]]
    _same_solve(host, dev)


def test_device_solve_return_index_edges(aoc):
    """Q7 edges through aoc_newton_solve: stop at kk = 0 (all-zero slot), at kk = 1 (the caller's initial
    iterate, also when it is not float32-valued), and no stop at all (last computed iterate)."""
    g, bp, op = _problem(aoc, "problem_step_T500")
    c = load_golden("g6_chain_step_T500")
    xx0 = c["xx_init"].copy(); xx0[:, 1:] += 1e-9
    xi, ui = np.stack([xx0, c["xx_init"]]), np.stack([c["uu_init"], c["uu_init"]])
    d0, d1 = c["descent"][0], c["descent"][1]
    for term, n_it in ((-1e30, 9), (0.5 * (d0 + d1), 9), (1e30, 1), (1e30, 2), (1e30, 4)):   # max_iters = 1: no iteration at all
        prm = aoc.make_params(max_iters=n_it, stepsize_0=1.0, armijo_maxiters=10, term_cond=term)
        s = aoc.NewtonBatchSolver(bp, 2, prm)
        s.set_initial(xi, ui)
        assert s.cur_is64
        host = s.solve(compact=False)
        s.set_initial(xi, ui)
        dev = s.solve_on_device(sync_every=1)
        _same_solve(host, dev)
        s.set_initial(xi, ui)
        _same_solve(host, s.solve_on_device(sync_every=1, x_star_f64=True))   # x_star stored as fp64
    assert dev["iters"].tolist() == [3, 3] and not dev["converged"].any()


def test_device_solve_with_repacking(aoc):
    """aoc_newton_solve re-packs the still-iterating trajectories into a smaller batch when at most half of
    the batch in flight is left (here twice: 2600 -> <=1300 -> ...); every per-trajectory result must equal
    the host loop without re-packing, bit for bit, including the trajectories that never stop."""
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 6000
    x0 = problems.perturbed_x0(pr, B, seed=77)
    prm = aoc.make_params(max_iters=40, stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    host = s.solve(compact=False)
    s.set_initial_from_x0(x0)
    dev = s.solve_on_device(sync_every=1)
    spread = np.sort(host["iters"])
    assert spread[B // 2] < spread[-1], "iteration counts must spread for re-packing to happen"
    _same_solve(host, dev)


def _solve_trace(aoc, s, **kw):
    """solve_on_device under aoc_solve_trace: (result, rows [part, kk, in flight, tiles, count read, ms])"""
    import ctypes as C
    from aircraftoptimalcontrol_amd._lib import lib
    rows = np.zeros((512, 6))
    lib().aoc_solve_trace(rows.ctypes.data_as(C.c_void_p), rows.shape[0])
    try:
        r = s.solve_on_device(**kw)
        n = lib().aoc_solve_trace_rows()
    finally:
        lib().aoc_solve_trace(None, 0)
    return r, rows[:n].copy()


def test_device_solve_retires_the_all_nan_fixed_point(aoc):
    """A diverged trajectory never passes the reference's stopping test (NaN >= -1e-6 is False, optcon.py:499) and
    iterates to max_iters on an iterate that no longer changes.  aoc_newton_solve retires it once one iteration has
    run on that all-NaN iterate, with exactly what the remaining iterations would have produced: from random x0 a few
    per cent of the batch diverge at the Hessian switch (kk = 9); the device solve must equal the host loop — which
    really runs every iteration — bit for bit (iterates, iteration counts = max_iters - 1, status flags, histories up to
    the last row), while its timeline shows that it stopped launching long before max_iters."""
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 2600
    x0 = problems.random_x0(B, seed=20260403)
    prm = aoc.make_params(max_iters=70, stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    host = s.solve(compact=False)
    s.set_initial_from_x0(x0)
    dev, rows = _solve_trace(aoc, s, sync_every=2)
    nan = np.isnan(host["uu_star"][:, :, :-1]).all((1, 2))
    assert 10 < nan.sum() < B // 4, "the case needs diverged trajectories"
    assert (host["iters"][nan] == prm.max_iters - 1).all() and not host["converged"][nan].any()
    assert host["last_kk"] == prm.max_iters - 2
    _same_solve(host, dev)
    launched = int(rows[rows[:, 1] >= 0, 1].max()) + 1
    assert launched < prm.max_iters - 10, "the retired trajectories should not keep the batch iterating (%d launched)" % launched


def test_device_solve_on_two_streams_equals_one(aoc, tuned):
    """aoc_newton_solve2 cuts a large batch in two halves on two HIP streams (each with its own generations and
    re-packing); with the threshold lowered so that 6000 trajectories split (47 + 47 tiles, the second half ragged), the
    results — iterates, counts, flags, histories — equal the one-stream solve bit for bit, and the timeline shows both
    parts at work.  Shared and per-trajectory reference curves (the second half reads its curves at a tile offset)."""
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(1.0, 2e-3)
    B = 6000
    x0 = problems.perturbed_x0(pr, B, seed=77)
    prm = aoc.make_params(max_iters=40, stepsize_0=1.0, armijo_maxiters=10)
    XR, UR = np.repeat(pr.xx_ref[None], B, 0), np.repeat(pr.uu_ref[None], B, 0)
    XR[B // 3:, 1] *= 0.9                       # two manoeuvres, the cut between the halves falls inside the second
    for bp in (aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt),
               aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, XR, UR, pr.dt)):
        s = aoc.NewtonBatchSolver(bp, B, prm)
        # (halves of 47 tiles would take the horizon cut in their Gauss-Newton iterations, the whole batch of 94 tiles
        # would not — aoc_tuning.bw_hcut, not bit-identical —: off, as for every split the library makes by itself,
        # whose halves have at least 1024 tiles)
        tuned(bw_hcut=0)
        s.set_initial_from_x0(x0)
        one = s.solve_on_device(sync_every=2, two_streams=False)
        tuned(solve_split_tiles=2, bw_hcut=0)
        s.set_initial_from_x0(x0)
        two, rows = _solve_trace(aoc, s, sync_every=2, two_streams=True)
        tuned(bw_hcut=0)
        assert sorted(set(rows[:, 0].astype(int))) == [0, 1]
        assert rows[rows[:, 0] == 0, 2].max() == 47 * 64 and rows[rows[:, 0] == 1, 2].max() == B - 47 * 64
        assert one["converged"].sum() > B // 2
        _same_solve(one, two)
        if bp.per_traj:   # lane b must keep reading curve b: a batch with its own curves is never re-packed (it was, in round 3)
            s.set_initial_from_x0(x0)
            _same_solve(s.solve(compact=False), one)


def test_device_summary_equals_the_host_summary(aoc):
    """aoc_summary (the five scalars the path's one collective reduces) against sharding.local_summary on the same
    per-trajectory arrays, non-finite costs included; two batches accumulated = the whole."""
    import torch
    from aircraftoptimalcontrol_amd import problems, sharding
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 1500
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(problems.random_x0(B, seed=20260403))
    for kk in range(11):
        s.iterate(kk)
    ref = sharding.local_summary(*s.summary_tensors()).cpu().numpy()
    got = s.summary().cpu().numpy()
    assert ref[4] > 0 and got[3] == B and got[4] == ref[4] and got[2] == ref[2]
    assert np.allclose(got[:2], ref[:2], rtol=1e-13, atol=0)
    assert np.array_equal(s.summary().cpu().numpy(), got)            # a fixed reduction order: the same bits again
    t = aoc.TwoStreamNewtonSolver(bp, B, prm)
    t.set_initial_from_x0(torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda())
    for kk in range(11):
        t.iterate(kk)
    both = t.summary().cpu().numpy()
    assert both[3] == B and both[4] == ref[4] and both[2] == ref[2] and np.allclose(both[:2], ref[:2], rtol=1e-13, atol=0)


def test_speculation_depth_does_not_change_results(aoc, tuned):
    """How many Armijo candidates ride along in the forward pass is a scheduling decision: 1, 2, 3 (one wavefront
    per tile or one per chain) and all 10 (tiny batches: several workgroups per tile, no trial round at all) must
    give bit-identical iterates, steps, trial counts and costs; aoc_default_nspec picks 10 for this batch."""
    from aircraftoptimalcontrol_amd import problems, _lib
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 300
    x0 = problems.random_x0(B, seed=31)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    assert _lib.lib().aoc_default_nspec(B, 10) == 11 and _lib.lib().aoc_default_nspec(200000, 10) == 2
    res = {}
    for ns, wl in ((1, 0), (2, 0), (3, 0), (10, 0), (11, 0), (7, 0), (2, 1), (3, 1)):
        tuned(nspec=ns, ls_worklist=wl)
        s = aoc.NewtonBatchSolver(bp, B, prm)
        assert s.n_spec == ns
        s.set_initial_from_x0(x0)
        hist = []
        for kk in range(4):
            s.iterate_timed(kk) if kk % 2 else s.iterate(kk)
            hist.append(s.scalars())
        res[ns, wl] = (hist, s.current())
    ha, (xa, ua) = res[2, 0]
    assert max(h["ntrials"].max() for h in ha) > 3, "some trajectories must back-track beyond the speculated steps"
    for ns in sorted(res):
        hb, (xb, ub) = res[ns]
        assert np.array_equal(xa, xb) and np.array_equal(ua, ub), ns
        for a, b in zip(ha, hb):
            for key in ("stepsize", "ntrials", "cost", "cost_new", "descent", "status"):
                assert np.array_equal(a[key], b[key]), (ns, key)


def test_split_kernels_equal_single_wavefront_kernels(aoc, tuned):
    """Small batches run the backward pass, the forward pass and the final rollout on several wavefronts per
    tile (k_backward2, k_backward4 / k_backward5, k_forward_split, k_ls_final_split); same operations on the same values, so the results
    must equal those of the one-wavefront-per-tile kernels bit for bit, across the full-Hessian switch."""
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 200
    x0 = problems.perturbed_x0(pr, B, seed=8)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    res = []
    for tiles, bw5 in ((0, 0), (512, 0), (512, 1)):     # one wavefront per tile | k_backward4 | k_backward5 in the Gauss-Newton iterations
        tuned(nspec=2, split_tiles=tiles, split_bw_tiles=tiles, ls_worklist=0, bw5=bw5)
        s = aoc.NewtonBatchSolver(bp, B, prm)
        s.set_initial_from_x0(x0)
        res.append((s.run_fixed(11), s.current()))
    (ha, (xa, ua)), (hb, (xb, ub)), (hc, (xc, uc)) = res
    assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True)
    assert np.array_equal(xa, xc, equal_nan=True) and np.array_equal(ua, uc, equal_nan=True)
    for a, c in zip(ha, hc):
        for key in ("stepsize", "ntrials", "cost", "cost_new", "descent", "status"):
            assert np.array_equal(a[key], c[key], equal_nan=True), key
    # the same for the tracking gains and the stored-state rollout (k_track_gains2/4, k_rollout_cost_split)
    tw = problems.tracking_weights()
    tp = aoc.BatchProblem(tw[0], tw[1], tw[2], pr.xx_ref, pr.uu_ref, pr.dt)
    outs = []
    for kn in (dict(split_tiles=0), dict(split_tiles=512, bw4_tiles=0), dict(split_tiles=512, bw4_tiles=256)):
        tuned(**kn)   # k_track_gains / k_track_gains2 / k_track_gains4
        g = aoc.lqr_tracking_batch(tp, xa[:70], ua[:70], 0.1)
        r = aoc.rollout_cost(bp, x0[:70], ua[:70], du=0.3 * ua[:70], alpha=np.full(70, 0.7))
        outs.append((g, r))
    ga, ra = outs[0]
    for gb, rb in outs[1:]:
        for a_, b_ in list(zip(ga, gb)) + list(zip(ra, rb)):
            assert np.array_equal(a_, b_, equal_nan=True)
    for a, b in zip(ha, hb):
        for key in ("stepsize", "ntrials", "cost", "cost_new", "descent", "status"):
            assert np.array_equal(a[key], b[key], equal_nan=True), key


def test_forward_pass_with_a_linearisation_wavefront_changes_nothing(aoc, tuned):
    """k_forward_lin (aoc_tuning.fw_lin): the nominal-point work of the small-batch forward pass on a wavefront of its
    own, the LQR wavefront reduced to the recursion proper.  Same operations on the same values: iterates, steps, trial
    counts, costs, descents and status flags must equal those of k_forward_split bit for bit — with 2, 3, 7 and all 10
    candidates riding along (1 to 5 workgroups per tile, spare trial wavefronts, candidate stores with and without a
    hint), on a ragged batch, across the Hessian switch, for an even and an odd number of stages and for a
    caller-supplied fp64 initial iterate."""
    from aircraftoptimalcontrol_amd import problems
    for T, B in ((500, 200), (333, 70)):
        pr = problems.step_maneuver(1.0, 1.0 / T)
        assert pr.T == T
        bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
        x0 = problems.random_x0(B, seed=23)
        prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
        for ns in (2, 3, 7, 10):
            res = []
            for lin in (0, 1):
                tuned(nspec=ns, fw_lin=lin, fw_duo=0)
                s = aoc.NewtonBatchSolver(bp, B, prm)
                s.set_initial_from_x0(x0)
                res.append((s.run_fixed(11 if ns == 10 else 4), s.current(), s.direction()))
            (ha, (xa, ua), da), (hb, (xb, ub), db) = res
            assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True), (T, ns)
            assert np.array_equal(da, db, equal_nan=True), (T, ns)
            for a, b in zip(ha, hb):
                for key in a:
                    assert np.array_equal(a[key], b[key], equal_nan=True), (T, ns, key)
    # a caller-supplied iterate with arbitrary fp64 samples is read as fp64 by the first iteration
    g = load_golden("g6_chain_step_T500")
    pg = load_golden("problem_step_T500")
    bp = aoc.BatchProblem(pg["QQt"], pg["RRt"], pg["QQT"], pg["xx_ref"], pg["uu_ref"], float(pg["dt"]))
    out = []
    for lin in (0, 1):
        tuned(fw_lin=lin, fw_duo=0)
        s = aoc.NewtonBatchSolver(bp, 3, aoc.make_params(stepsize_0=1.0, armijo_maxiters=10))
        xi = np.repeat(g["xx_init"][None], 3, 0) * (1 + 1e-13 * np.arange(3)[:, None, None])
        s.set_initial(xi, np.repeat(g["uu_init"][None], 3, 0))
        s.iterate(0)
        out.append((s.scalars(), s.current()))
    for key in out[0][0]:
        assert np.array_equal(out[0][0][key], out[1][0][key], equal_nan=True), key
    assert np.array_equal(out[0][1][0], out[1][1][0]) and np.array_equal(out[0][1][1], out[1][1][1])


def test_forward_pass_on_seven_roles_changes_nothing(aoc, tuned):
    """k_forward_duo (aoc_tuning.fw_duo; round 5): the forward pass of tiny batches with the stage cut into seven roles
    (two nominal-point producers, the LQR recursion, two candidates as state chain | cost).  Same operations on the same
    values as k_forward_split: iterates, directions, steps, trial counts, costs, descents and status flags bit-identical —
    with 4, 7, 10 and 11 candidates riding along (spare candidate of the last group, candidate stores with and without a
    hint, the step of an exhausted search as candidate 10), on a ragged batch, across the Hessian switch, for even and odd
    numbers of stages down to T = 3, with per-trajectory reference curves, with dense weights and for a caller-supplied
    fp64 initial iterate."""
    from aircraftoptimalcontrol_amd import problems
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)

    def both(bp, B, x0, n_it, ns, init=None):
        res = []
        for duo in (0, 1):
            tuned(nspec=ns, fw_duo=duo)
            s = aoc.NewtonBatchSolver(bp, B, prm)
            if init is None:
                s.set_initial_from_x0(x0)
            else:
                s.set_initial(*init)
            res.append((s.run_fixed(n_it), s.current(), s.direction()))
        (ha, (xa, ua), da), (hb, (xb, ub), db) = res
        assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True), (B, ns)
        assert np.array_equal(da, db, equal_nan=True), (B, ns)
        for a, b in zip(ha, hb):
            for key in a:
                assert np.array_equal(a[key], b[key], equal_nan=True), (B, ns, key)

    for T, B in ((500, 200), (333, 70)):
        pr = problems.step_maneuver(1.0, 1.0 / T)
        bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
        x0 = problems.random_x0(B, seed=23)
        for ns in (4, 7, 10, 11):
            both(bp, B, x0, 11 if ns >= 10 else 4, ns)
    # horizons shorter than the load rings and the hand-off blocks
    pr = problems.step_maneuver(1.0, 2e-3)
    for T in (3, 4, 5, 6, 9):
        bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref[:, :T], pr.uu_ref[:, :T], pr.dt)
        both(bp, 70, problems.perturbed_x0(pr, 70, seed=T), 3, 11)
    # one reference curve per trajectory; dense weights
    B = 130
    XR, UR = np.repeat(pr.xx_ref[None], B, 0), np.repeat(pr.uu_ref[None], B, 0)
    XR[B // 2:, 1] *= 0.9
    both(aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, XR, UR, pr.dt), B, problems.perturbed_x0(pr, B, seed=4), 4, 11)
    rng = np.random.default_rng(0)
    Q, QT, R = pr.QQt.copy(), pr.QQT.copy(), pr.RRt.copy()
    for M, sc in ((Q, 1e-3), (QT, 1e-2)):
        A = rng.normal(size=(6, 6)) * sc
        M += A @ A.T
    R += np.array([[0.0, 2e-7], [2e-7, 0.0]])
    both(aoc.BatchProblem(Q, R, QT, pr.xx_ref, pr.uu_ref, pr.dt), 70, problems.perturbed_x0(pr, 70, seed=5), 4, 11)
    # a caller-supplied iterate with arbitrary fp64 samples is read as fp64 by the first iteration
    g = load_golden("g6_chain_step_T500")
    pg = load_golden("problem_step_T500")
    bpg = aoc.BatchProblem(pg["QQt"], pg["RRt"], pg["QQT"], pg["xx_ref"], pg["uu_ref"], float(pg["dt"]))
    xi = np.repeat(g["xx_init"][None], 3, 0) * (1 + 1e-13 * np.arange(3)[:, None, None])
    both(bpg, 3, None, 2, 11, init=(xi, np.repeat(g["uu_init"][None], 3, 0)))


def test_forward_state_recomputation_does_not_change_results(aoc, tuned):
    """aoc_forward re-computes the nominal states x_{t+1} = step(x_t, u_t) instead of reading them when the iterate is a
    rollout the library wrote (aoc_problem.x_is_rollout; one-wavefront-per-tile kernel): same operations as the rollout
    that stored them, so every result must be bit-identical to the reading variant — for 1, 2 and 3 speculated trials,
    across the Hessian switch; and a caller-supplied iterate (not a rollout) must still be READ."""
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 200
    x0 = problems.random_x0(B, seed=17)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    for ns in (1, 2, 3):
        res = []
        for rc in (0, 1):
            tuned(nspec=ns, split_tiles=0, split_bw_tiles=0, fw_recompute=rc)
            s = aoc.NewtonBatchSolver(bp, B, prm)
            s.set_initial_from_x0(x0)
            assert s.cur_rollout
            res.append((s.run_fixed(11), s.current()))
        (ha, (xa, ua)), (hb, (xb, ub)) = res
        assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True), ns
        for a, b in zip(ha, hb):
            for key in a:
                assert np.array_equal(a[key], b[key], equal_nan=True), (ns, key)
    # an initial iterate that is NOT the rollout of its inputs (float32-valued states shifted by one float32 ulp)
    tuned(nspec=2, split_tiles=0, split_bw_tiles=0, fw_recompute=1)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    xi, ui = s.current()
    xi2 = xi.copy()
    xi2[:, :, 1:] = np.nextafter(xi[:, :, 1:].astype(np.float32), np.float32(np.inf)).astype(np.float64)
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    s.set_initial(xi2, ui)
    assert not s.cur_rollout and not s.cur_is64
    s.iterate(0)
    sc = s.scalars()
    xn, un = s.current()
    for b in (0, 77, 199):
        r = orc.newton_iterate(op, orc.params(), 0, xi2[b], ui[b], xi2[b][:, 0])
        assert r["stepsize"] == sc["stepsize"][b] and abs(r["descent"] - sc["descent"][b]) <= 1e-8 * abs(r["descent"])
        assert rel_err(un[b], r["uu"], 1e-3) < 1e-8


def test_stored_candidates_do_not_change_results(aoc, tuned):
    """Small batches: every Armijo candidate rides along in the forward pass, the trial wavefronts keep the trajectories
    they roll out, and the update of a tile whose trajectories all accepted one of them is a copy instead of a rollout
    (aoc_tuning.store_candidates).  Against the rollout update: bit-identical iterates, costs and flags over 22
    fixed iterations — the late ones exhaust line searches, whose tiles must fall back to the rollout."""
    from aircraftoptimalcontrol_amd import problems, _lib
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 330
    x0 = problems.perturbed_x0(pr, B, seed=14)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    for ns in (10,):
        res = []
        for keep in (0, 1):
            tuned(nspec=ns, store_candidates=keep)
            s = aoc.NewtonBatchSolver(bp, B, prm)
            s.set_initial_from_x0(x0)
            hist = []
            for kk in range(22):
                s.iterate_timed(kk) if kk % 3 == 2 else s.iterate(kk)
                hist.append(s.scalars())
            res.append((hist, s.current()))
        (ha, (xa, ua)), (hb, (xb, ub)) = res
        assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True), ns
        for a, b in zip(ha, hb):
            for key in a:
                assert np.array_equal(a[key], b[key], equal_nan=True), (ns, key)
        assert any((h["status"] & _lib.ST_ARMIJO_EXH).any() for h in ha), "the run should contain exhausted searches"


def test_four_wavefront_backward_equals_the_others(aoc, tuned):
    """Tiny batches run the Gauss-Newton backward pass on four wavefronts per tile (k_backward4: one producer, the Riccati
    columns {0,1,2}, {3,4}, {5} + affine terms on three consumers, constants pinned in VGPRs).  Against the
    two-wavefront and the one-wavefront kernel: gains, direction, descent and every iterate over 12 iterations bit for
    bit (the full-Hessian iterations run k_backward2 in every variant), incl. a golden Gauss-Newton case."""
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 150
    x0 = problems.random_x0(B, seed=23)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    variants = (dict(bw4_tiles=256), dict(bw4_tiles=0), dict(bw4_tiles=0, split_bw_tiles=0))
    res = []
    for kn in variants:
        tuned(**kn)
        s = aoc.NewtonBatchSolver(bp, B, prm)
        s.set_initial_from_x0(x0)
        res.append((s.run_fixed(12), s.current()))
    ha, (xa, ua) = res[0]
    assert max(int(h["status"].max()) for h in ha) & 8, "some stage should have been regularised after the Hessian switch"
    for hb, (xb, ub) in res[1:]:
        assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True)
        for a, b in zip(ha, hb):
            for key in a:
                assert np.array_equal(a[key], b[key], equal_nan=True), key
    _, bp2, _ = _problem(aoc, "problem_step_T1000")
    for case, fh in (("g3_lqr_a_gn_init_T1000", 0), ("g3_lqr_b_full_init_T1000", 1)):   # b: 18 regularised stages
        g = load_golden(case)
        outs = []
        for kn in variants:
            tuned(**kn)
            s_ = aoc.NewtonBatchSolver(bp2, 1, prm)      # Gauss-Newton through aoc_newton_iterate: no costate asked for
            xi = g["xx"][None].copy()
            s_.set_initial(xi, g["uu"][None])
            s_.iterate(9 if fh else 0)
            outs.append((s_.scalars(), s_.current(), aoc.backward_forward(bp2, g["xx"][None], g["uu"][None], fh)))
        for sc, (xx, uu), o in outs[1:]:
            for key in sc:
                assert np.array_equal(outs[0][0][key], sc[key]), (case, key)
            assert np.array_equal(outs[0][1][0], xx) and np.array_equal(outs[0][1][1], uu), case
            for key in ("KK", "du", "descent", "lmbd0", "J_trial0", "status"):
                assert np.array_equal(outs[0][2][key], o[key]), (case, key)
