"""The horizon cut (aoc_tuning.bw_hcut / track_hcut; k_bw_hcut, k_track_hcut_*): for batches that leave most SIMDs idle the
Riccati recursion of a Gauss-Newton backward pass / of the tracking gains is evaluated in S horizon segments in parallel
(each segment as ONE map of the value function at its end), a short serial sweep hands the boundary values down, and
every segment then computes its gains from its own boundary.  Another order of the same arithmetic: NOT bit-identical
to the sequential kernels.  The acceptance gates: gains within 1e-8 of their scale (measured 1e-14),
Armijo steps and trial counts identical, on the golden chains (tests/test_gpu_parity.py and test_gpu_drivers.py run with
the cut by default: a single trajectory is one tile) and on the 4096-trajectory sweeps (tests/test_gpu_sweep.py: 64 tiles)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _gains(s):
    from aircraftoptimalcontrol_amd import batch as aoc
    Kt = s._carve()[0]
    return aoc.unpack_gains(Kt.view(s.nt, s.T, 14, 64), s.B).cpu().numpy()


@pytest.mark.parametrize("B", [1, 70, 1000, 4096])
def test_cut_backward_pass_equals_the_sequential_one(tuned, B):
    """One Gauss-Newton iteration with the backward pass cut in 4, 8, 16 segments (and whatever the default picks)
    against the sequential multi-wavefront kernels: every gain within 1e-12 of the largest gain of its column (measured
    1e-14), direction and descent within 1e-11, the same Armijo step and trial count for every trajectory, the same new
    float32 states; then nine iterations free-running: identical step and trial histories."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.random_x0(B, seed=11) if B != 1000 else problems.perturbed_x0(pr, B, seed=11)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    out = {}
    for S in (0, 4, 8, 16, -1):
        if S > 0 and S * s.nt > 1024:
            continue
        tuned(bw_hcut=S)
        s.set_initial_from_x0(x0)
        s.iterate(0)
        K = _gains(s)
        du, sc, (xx, uu) = s.direction(), s.scalars(), s.current()
        hist = [(sc["stepsize"].copy(), sc["ntrials"].copy())]
        for kk in range(1, 9):
            s.iterate(kk)
            sc2 = s.scalars()
            hist.append((sc2["stepsize"].copy(), sc2["ntrials"].copy()))
        out[S] = (K, du, sc, xx, uu, hist, s.current())
    K0, du0, sc0, xx0, uu0, hist0, (xf0, uf0) = out[0]
    kscale = np.abs(K0).max(axis=(0, 3), keepdims=True)
    for S, (K, du, sc, xx, uu, hist, (xf, uf)) in out.items():
        if S == 0:
            continue
        assert (np.abs(K - K0) / kscale).max() < 1e-12, S
        assert np.abs(du - du0).max() <= 1e-11 * np.abs(du0).max(), S
        assert np.allclose(sc["descent"], sc0["descent"], rtol=1e-11, atol=0), S
        assert np.array_equal(sc["stepsize"], sc0["stepsize"]) and np.array_equal(sc["ntrials"], sc0["ntrials"]), S
        assert np.array_equal(sc["status"], sc0["status"]), S
        assert np.array_equal(xx, xx0), S
        for (a, b), (c, d) in zip(hist, hist0):
            assert np.array_equal(a, c) and np.array_equal(b, d), S
        same = sum(np.array_equal(xf[b], xf0[b]) for b in range(B))
        assert same >= B - max(1, B // 500), (S, same)          # a float32 rounding flip is possible, none was seen
        chan = (np.abs(uf - uf0).max(2) / np.maximum(np.abs(uf0).max(2), 1e-3)).max(1)
        # nine iterations free-running: 1e-14 grows along the ill-conditioned trajectories of the random starts (measured
        # 1.8e-8 on one of 4096; 2e-14 from perturbed starts)
        assert chan.max() < 1e-6 and np.median(chan) < 1e-11, (S, chan.max(), np.median(chan))
    if B <= 4096:   # the default cuts batches of at most 64 tiles: not the sequential result bit for bit
        assert not np.array_equal(out[-1][0], K0) and np.array_equal(out[-1][0], out[16][0])


def test_cut_tracking_gains_equal_the_sequential_ones(tuned):
    """aoc_mpc_step's tracking gains (lqr_tracking.py:268-276) with the horizon cut against the sequential kernels: gains
    within 1e-12 of their scale, the applied input within 1e-12, the plant state identical."""
    from aircraftoptimalcontrol_amd import mpc, problems
    T, steps, B = 500, 30, 200
    L = T + steps + 10
    full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
    pr = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
    res = {}
    for S in (0, 8, 16, -1):
        tuned(track_hcut=S, bw_hcut=0)
        rh = mpc.RecedingHorizon(pr, problems.tracking_weights(), B, T, n_newton=2, sigma=np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002]))
        rh.start(problems.perturbed_x0(pr, B, seed=1), cold_iters=6)
        outs = [rh.step() for _ in range(3)]
        res[S] = (rh.Kg.cpu().numpy().copy(), outs)
    K0, o0 = res[0]
    scale = np.abs(K0).max(axis=(0, 1, 3), keepdims=True)
    for S in (8, 16, -1):
        K, o = res[S]
        assert (np.abs(K - K0) / scale).max() < 1e-12, S
        for a, b in zip(o, o0):
            assert np.abs(a["u_applied"] - b["u_applied"]).max() <= 1e-12 * np.abs(b["u_applied"]).max(), S
            assert np.array_equal(a["x_true"], b["x_true"]), S
            assert np.allclose(a["cost"], b["cost"], rtol=1e-12, atol=0), S
    assert np.array_equal(res[-1][0], res[16][0]) and not np.array_equal(res[16][0], K0)


def test_full_hessian_iterations_and_large_batches_are_not_cut(tuned):
    """kk > 8 (costate and Hessian terms) and batches above 64 tiles keep the sequential kernels whatever the knob says:
    bit-identical results with the cut forced on and off."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    for B, kks in ((300, (9, 10)), (65 * 64 + 1, (0, 1))):
        x0 = problems.perturbed_x0(pr, B, seed=3)
        s = aoc.NewtonBatchSolver(bp, B, prm)
        res = []
        for S in (0, -1):
            tuned(bw_hcut=S)
            s.set_initial_from_x0(x0)
            for kk in kks:
                s.iterate(kk)
            res.append((s.current(), s.scalars()))
        (xa, ua), sa = res[0]
        (xb, ub), sb = res[1]
        assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True), B
        for key in sa:
            assert np.array_equal(sa[key], sb[key], equal_nan=(sa[key].dtype.kind == "f")), (B, key)
