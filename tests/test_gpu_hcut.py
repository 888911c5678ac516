"""The horizon cut (aoc_tuning.bw_hcut / track_hcut; k_bw_hcut, k_track_hcut_*): for batches that leave most SIMDs idle the
Riccati recursion of a Gauss-Newton backward pass / of the tracking gains is evaluated in S horizon segments in parallel
(each segment as ONE map of the value function at its end), a short serial sweep hands the boundary values down, and
every segment then computes its gains from its own boundary.  Another order of the same arithmetic: NOT bit-identical
to the sequential kernels.  The acceptance gates: gains within 1e-8 of their scale (measured 1e-14),
Armijo steps and trial counts identical, on the golden chains (tests/test_gpu_parity.py and test_gpu_drivers.py run with
the cut by default: a single trajectory is one tile) and on the 4096-trajectory sweeps (tests/test_gpu_sweep.py: 64 tiles)."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _same_float32_states(xa, xb, frac=1e-5):
    """The cut moves inputs in their 14th digit; a float32 state stays what it was unless that tips a rounding (measured:
    0 or 1 element of 1.9 million, one ulp — tools/probes/cut_state_flips.py).  Sample 0 is the caller's fp64 x0."""
    if not np.array_equal(xa[:, :, 0], xb[:, :, 0]):
        return False
    a, b = xa[:, :, 1:].astype(np.float32), xb[:, :, 1:].astype(np.float32)
    ulps = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    return ulps.max() <= 1 and np.count_nonzero(ulps) <= max(1, frac * ulps.size)


def _gains(s):
    from aircraftoptimalcontrol_amd import batch as aoc
    Kt = s._carve()[0]
    # (sample T-1 of K~ is never written — aoc.h: "sample T-1 unused" — and holds whatever the allocator's block held before)
    return aoc.unpack_gains(Kt.view(s.nt, s.T, 14, 64), s.B).cpu().numpy()[..., :-1]


@pytest.mark.parametrize("B", [1, 70, 1000, 4096, 8000])
def test_cut_backward_pass_equals_the_sequential_one(tuned, B):
    """One Gauss-Newton iteration with the backward pass cut in 4, 8, 16 segments (and whatever the default picks)
    against the sequential multi-wavefront kernels: every gain within 1e-12 of the largest gain of its column (measured
    1e-14), direction and descent within 1e-11, the same Armijo step and trial count for every trajectory, the same new
    float32 states (but for a rounding tipped once in a million); then nine iterations free-running: identical step and
    trial histories."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.random_x0(B, seed=11) if B != 1000 else problems.perturbed_x0(pr, B, seed=11)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    out = {}
    # (segments, maps of neighbouring segments composed pairwise before the chain: aoc_tuning.hcut_pairs); odd numbers of
    # segments leave one plain hop behind the composed ones
    for S in (0, 4, 8, 16, -1, (16, 0), (8, 2), (5, 2), (7, 2), (3, 2), (2, 2)):
        S, pairs = S if isinstance(S, tuple) else (S, 1)
        if S > 0 and S * s.nt > 1024:
            continue
        tuned(bw_hcut=S, hcut_pairs=pairs)
        key = S if pairs == 1 else (S, pairs)
        s.set_initial_from_x0(x0)
        s.iterate(0)
        K = _gains(s)
        du, sc, (xx, uu) = s.direction(), s.scalars(), s.current()
        hist = [(sc["stepsize"].copy(), sc["ntrials"].copy())]
        for kk in range(1, 9):
            s.iterate(kk)
            sc2 = s.scalars()
            hist.append((sc2["stepsize"].copy(), sc2["ntrials"].copy()))
        out[key] = (K, du, sc, xx, uu, hist, s.current())
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
        assert _same_float32_states(xx, xx0), S
        for (a, b), (c, d) in zip(hist, hist0):
            assert np.array_equal(a, c) and np.array_equal(b, d), S
        same = sum(np.array_equal(xf[b], xf0[b]) for b in range(B))
        assert same >= B - max(1, B // 500), (S, same)          # a float32 rounding flip is possible, none was seen
        chan = (np.abs(uf - uf0).max(2) / np.maximum(np.abs(uf0).max(2), 1e-3)).max(1)
        # nine iterations free-running: 1e-14 grows along the ill-conditioned trajectories of the random starts (measured
        # 1.8e-8 on one of 4096; 2e-14 from perturbed starts)
        assert chan.max() < 1e-6 and np.median(chan) < 1e-11, (S, chan.max(), np.median(chan))
    # the default cuts batches of at most 64 tiles in 16 segments, up to 128 tiles in 8: not the sequential result bit for bit
    assert not np.array_equal(out[-1][0], K0) and np.array_equal(out[-1][0], out[16 if B <= 4096 else 8][0])
    if (16, 0) in out:      # composing the maps is another order of the arithmetic again
        assert not np.array_equal(out[(16, 0)][0], out[16][0])


def test_cut_tracking_gains_equal_the_sequential_ones(tuned):
    """aoc_mpc_step's tracking gains (lqr_tracking.py:268-276) with the horizon cut against the sequential kernels: gains
    within 1e-12 of their scale, the applied input within 1e-12, the plant state identical (or one float32 rounding apart)."""
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
            assert np.array_equal(a["x_true"], b["x_true"]) or np.abs(a["x_true"] - b["x_true"]).max() <= 2e-7 * np.abs(b["x_true"]).max(), S
            assert np.allclose(a["cost"], b["cost"], rtol=1e-12, atol=0), S
    assert np.array_equal(res[-1][0], res[16][0]) and not np.array_equal(res[16][0], K0)


@pytest.mark.parametrize("B", [70, 4096])
def test_six_wavefront_chain_equals_the_one_wavefront_chain(tuned, B):
    """k_hcut_chain6 (round 5: the boundary hops of the horizon cut on six wavefronts per tile, one column of the dense 6x6
    algebra each) against k_track_hcut_chain: the same expressions entry by entry, so gains, directions, new iterates and
    every scalar are bit-identical — for the Gauss-Newton backward pass and for the tracking gains of aoc_mpc_step."""
    from aircraftoptimalcontrol_amd import batch as aoc, mpc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.random_x0(B, seed=5)
    s = aoc.NewtonBatchSolver(bp, B, aoc.make_params(stepsize_0=1.0, armijo_maxiters=10))
    res = []
    for c6 in (0, 1):
        tuned(hcut_chain6=c6, hcut_pairs=0)
        s.set_initial_from_x0(x0)
        s.iterate(0)
        K = _gains(s)
        s.iterate(1)
        res.append((K, s.current(), s.scalars()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1][0], res[1][1][0]) and np.array_equal(res[0][1][1], res[1][1][1])
    for key in res[0][2]:
        assert np.array_equal(res[0][2][key], res[1][2][key]), key
    if B == 70:
        T, L = 200, 240
        full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
        prm_ = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
        outs = []
        for c6 in (0, 1):
            tuned(hcut_chain6=c6, hcut_pairs=0)
            rh = mpc.RecedingHorizon(prm_, problems.tracking_weights(), B, T, n_newton=2, sigma=np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002]))
            rh.start(problems.perturbed_x0(prm_, B, seed=1), cold_iters=4)
            o = [rh.step() for _ in range(3)]
            outs.append((rh.Kg.cpu().numpy().copy(), o))
        assert np.array_equal(outs[0][0], outs[1][0])
        for a, b in zip(outs[0][1], outs[1][1]):
            for key in a:
                assert np.array_equal(a[key], b[key]), key


@pytest.mark.parametrize("B", [70, 1024, 4096])
def test_segment_kernels_on_several_wavefronts_change_nothing(tuned, B):
    """aoc_tuning.hcut_waves (round 5): the segment kernels of the horizon cut with a map stage on three wavefronts
    (linearisation | H half | Phi half: k_track_hcut_map3, k_bw_hcut_map3) and the recursions on two (k_*_hcut_gains2),
    against the one-wavefront kernels: same expressions, the values that cross between wavefronts are opaque to the
    compiler in the one-wavefront kernels, so gains, iterates and every scalar are bit-identical — Gauss-Newton and
    full-Hessian backward passes of the Newton iteration, and the tracking gains of aoc_mpc_step.  (4096 trajectories:
    64 tiles, more wavefronts than SIMDs, forced with hcut_waves = 2.)"""
    from aircraftoptimalcontrol_amd import batch as aoc, mpc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.perturbed_x0(pr, B, seed=5)
    s = aoc.NewtonBatchSolver(bp, B, aoc.make_params(stepsize_0=1.0, armijo_maxiters=10))
    res = []
    for w in (0, 2):
        tuned(hcut_waves=w)
        s.set_initial_from_x0(x0)
        out = []
        for kk in (0, 1, 9, 10):      # two Gauss-Newton iterations, two with the full Hessian
            s.iterate(kk)
            out.append((_gains(s), s.current(), s.scalars()))
        res.append(out)
    for a, b in zip(*res):
        assert np.array_equal(a[0], b[0], equal_nan=True)
        assert np.array_equal(a[1][0], b[1][0], equal_nan=True) and np.array_equal(a[1][1], b[1][1], equal_nan=True)
        for key in a[2]:
            assert np.array_equal(a[2][key], b[2][key], equal_nan=True), key
    if B <= 1024:
        T, L = 200, 240
        full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
        prm_ = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
        outs = []
        for w in (0, 2):
            tuned(hcut_waves=w)
            rh = mpc.RecedingHorizon(prm_, problems.tracking_weights(), B, T, n_newton=2, sigma=np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002]))
            rh.start(problems.perturbed_x0(prm_, B, seed=1), cold_iters=4)
            o = [rh.step() for _ in range(3)]
            outs.append((rh.Kg.cpu().numpy().copy(), o))
        assert np.array_equal(outs[0][0], outs[1][0])
        for a, b in zip(outs[0][1], outs[1][1]):
            for key in a:
                assert np.array_equal(a[key], b[key]), key


def test_segment_kernels_on_several_wavefronts_other_variants(tuned):
    """The same bit-identity for the other instantiations of the multi-wavefront segment kernels: dense weights with one
    reference curve per trajectory, the shortest horizon the cut takes (T - 1 = 4 segments' worth: 65 stages for 16) and an
    odd one, and a caller-supplied fp64 iterate (read as fp64 by the first backward pass) — run_fixed across the Hessian
    switch, every history entry, the iterate and the direction."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    B = 130
    rng = np.random.default_rng(0)
    Q, QT, R = pr.QQt.copy(), pr.QQT.copy(), pr.RRt.copy()
    for M, sc in ((Q, 1e-3), (QT, 1e-2)):
        A = rng.normal(size=(6, 6)) * sc
        M += A @ A.T
    R += np.array([[0.0, 2e-7], [2e-7, 0.0]])
    XR, UR = np.repeat(pr.xx_ref[None], B, 0), np.repeat(pr.uu_ref[None], B, 0)
    XR[B // 2:, 1] *= 0.9

    def both(bp, n_it, x0=None, init=None):
        res = []
        for w in (0, 2):
            tuned(hcut_waves=w)
            s = aoc.NewtonBatchSolver(bp, B, prm)
            if init is None:
                s.set_initial_from_x0(x0)
            else:
                s.set_initial(*init)
            res.append((s.run_fixed(n_it), s.current(), s.direction(), _gains(s)))
        (ha, (xa, ua), da, ka), (hb, (xb, ub), db, kb) = res
        assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True)
        assert np.array_equal(da, db, equal_nan=True) and np.array_equal(ka, kb, equal_nan=True)
        for a, b in zip(ha, hb):
            for key in a:
                assert np.array_equal(a[key], b[key], equal_nan=True), key

    x0 = problems.perturbed_x0(pr, B, seed=4)
    for T in (66, 133, 500):
        both(aoc.BatchProblem(Q, R, QT, XR[:, :, :T], UR[:, :, :T], pr.dt), 11, x0=x0)
    g = load_golden("g6_chain_step_T500")
    pg = load_golden("problem_step_T500")
    bpg = aoc.BatchProblem(pg["QQt"], pg["RRt"], pg["QQT"], pg["xx_ref"], pg["uu_ref"], float(pg["dt"]))
    xi = np.repeat(g["xx_init"][None], B, 0) * (1 + 1e-13 * np.arange(B)[:, None, None])
    both(bpg, 2, init=(xi, np.repeat(g["uu_init"][None], B, 0)))


@pytest.mark.parametrize("level", [1, 2])
def test_full_hessian_cut_against_the_sequential_kernels(tuned, level):
    """The horizon cut of FULL-HESSIAN backward passes (round 5: costate maps, value-function maps with the Hessian terms,
    chain, gains, and the sequential kernel for the lanes it does not trust) against the sequential kernels alone
    (bw_hcut_full = 0), iteration kk = 9 from the same iterate and five more free-running.
    Lanes whose M stays positive definite (all but a handful of the perturbed starts, 90 % of the random ones): gains
    within 1e-10 of their column's scale, direction and descent 1e-9, Armijo steps, trial counts, status flags and new
    float32 states identical, histories identical over kk = 9..14 (perturbed starts).
    Lanes that regularise (M indefinite: the reference regularises the gains, not the Riccati recursion, Q3):
      level 1: recomputed by the sequential kernel — gains bit-identical, per lane, whatever their tile-mates do;
      level 2: left to the cut unless M is singular / ill-conditioned or the value function not a number — the same lanes
               regularise and the same diverge; their gains are compared at the scale the regime allows (reported)."""
    import json, os
    from conftest import ROOT
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    rec = {}
    for name, B in (("perturbed", 300), ("perturbed", 4096), ("random", 4096)):
        x0 = problems.perturbed_x0(pr, B, seed=3) if name == "perturbed" else problems.random_x0(B, seed=11)
        s = aoc.NewtonBatchSolver(bp, B, prm)
        out = {}
        for full in (0, level):
            tuned(bw_hcut_full=full)
            s.set_initial_from_x0(x0)
            for kk in range(9):
                s.iterate(kk)                    # Gauss-Newton iterations: the same kernels either way
            start = s.current()
            s.status.zero_()
            s.iterate(9)
            K, du, sc, it = _gains(s), s.direction(), s.scalars(), s.current()
            hist = []
            for kk in range(10, 15):
                s.iterate(kk)
                sc2 = s.scalars()
                hist.append((sc2["stepsize"].copy(), sc2["ntrials"].copy()))
            out[full] = (start, K, du, sc, it, hist)
        (st0, K0, du0, sc0, it0, h0), (st1, K1, du1, sc1, it1, h1) = out[0], out[level]
        assert np.array_equal(st0[0], st1[0], equal_nan=True) and np.array_equal(st0[1], st1[1], equal_nan=True)
        assert np.array_equal(sc0["status"], sc1["status"]), (name, B, int((sc0["status"] != sc1["status"]).sum()))
        fin0, fin1 = np.isfinite(sc0["cost_new"]), np.isfinite(sc1["cost_new"])
        assert np.array_equal(fin0, fin1), (name, B, int((fin0 != fin1).sum()))
        flagged = (sc0["status"] & 12) != 0               # regularised or singular somewhere along the horizon
        okK = np.isfinite(K0).all((1, 2, 3)) & np.isfinite(K1).all((1, 2, 3))
        clean = okK & ~flagged
        kscale = np.abs(K0[clean]).max(axis=(0, 3), keepdims=True)
        kerr = (np.abs(K1[clean] - K0[clean]) / kscale).max()
        same_lanes = int(sum(np.array_equal(K0[b], K1[b]) for b in np.nonzero(clean)[0]))
        assert same_lanes <= clean.sum() // 10 and kerr < 1e-10, (name, B, same_lanes, kerr)    # the clean lanes really went through the cut
        dscale = np.abs(du0[clean]).max()
        assert np.abs(du1[clean] - du0[clean]).max() <= 1e-9 * dscale
        assert np.allclose(sc1["descent"][clean], sc0["descent"][clean], rtol=1e-9, atol=0)
        assert np.array_equal(sc1["stepsize"][clean], sc0["stepsize"][clean]) and np.array_equal(sc1["ntrials"][clean], sc0["ntrials"][clean])
        assert _same_float32_states(it1[0][clean], it0[0][clean])
        r = {"B": B, "lanes_regularised_or_singular": int(flagged.sum()), "gain_error_clean_lanes": float(kerr)}
        if flagged.any():
            fl = np.nonzero(flagged & okK)[0]
            if level == 1:
                for b in fl:
                    assert np.array_equal(K0[b], K1[b]), (name, B, int(b))
                assert np.array_equal(sc1["stepsize"][flagged], sc0["stepsize"][flagged])
            else:
                ks = np.abs(K0[fl]).max(axis=(0, 3), keepdims=True)
                e = (np.abs(K1[fl] - K0[fl]) / ks).max(axis=(1, 2, 3))
                r.update(gain_error_regularised_lanes_max=float(e.max()), gain_error_regularised_lanes_median=float(np.median(e)),
                         armijo_verdicts_differing=int((sc1["stepsize"][flagged] != sc0["stepsize"][flagged]).sum()))
                assert np.median(e) < 1e-6 and (sc1["stepsize"][flagged] != sc0["stepsize"][flagged]).sum() <= max(2, flagged.sum() // 50), r
        if name == "perturbed":
            same_h = sum(np.array_equal(a, c) and np.array_equal(b, d) for (a, b), (c, d) in zip(h1, h0))
            assert same_h == len(h0) or level == 2, (name, B, same_h)
        else:
            assert flagged.sum() > B // 50                              # the regime is really exercised
        rec["%s_%d" % (name, B)] = r
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", "hcut_full_level%d.json" % level), "w"), indent=1)


def test_large_batches_are_not_cut(tuned):
    """Batches above 128 tiles keep the sequential kernels by default (above 64 tiles for the full-Hessian iterations), and
    with bw_hcut_full = 0 so do the full-Hessian iterations (kk > 8) of small ones: bit-identical results with the cut
    switched off (bw_hcut = 0) and left to the default (-1)."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    for B, kks, full in ((300, (9, 10), 0), (128 * 64 + 1, (0, 1), 2), (64 * 64 + 1, (9, 10), 2)):
        x0 = problems.perturbed_x0(pr, B, seed=3)
        s = aoc.NewtonBatchSolver(bp, B, prm)
        res = []
        for S in (0, -1):
            tuned(bw_hcut=S, bw_hcut_full=full)
            s.set_initial_from_x0(x0)
            for kk in kks:
                s.iterate(kk)
            res.append((s.current(), s.scalars()))
        (xa, ua), sa = res[0]
        (xb, ub), sb = res[1]
        assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True), B
        for key in sa:
            assert np.array_equal(sa[key], sb[key], equal_nan=(sa[key].dtype.kind == "f")), (B, key)
