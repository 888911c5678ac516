"""Edge cases of the HIP path against the oracle: tiny horizons (shorter than the prefetch rings), ragged
and single-trajectory batches, dense (non-diagonal) weights, the reference's constructor defaults
(stepsize_0 = 1e-2, armijo_maxiters = 20), exhaustion of the line search, degenerate inputs."""
import numpy as np
import pytest

from conftest import load_golden, rel_err
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def aoc():
    from aircraftoptimalcontrol_amd import batch
    return batch


def _setup(aoc, T, dense=False, seed=0):
    from aircraftoptimalcontrol_amd import problems
    pr = problems.step_maneuver(tf=1.0, dt=1.0 / 500)
    Q, R, QT = pr.QQt.copy(), pr.RRt.copy(), pr.QQT.copy()
    if dense:
        rng = np.random.default_rng(seed)
        for M, n, sc in ((Q, 6, 1e-3), (QT, 6, 1e-2)):
            A = rng.normal(size=(n, n)) * sc
            M += A @ A.T
        R += np.array([[0.0, 2e-7], [2e-7, 0.0]])
    xr, ur = pr.xx_ref[:, :T], pr.uu_ref[:, :T]
    bp = aoc.BatchProblem(Q, R, QT, xr, ur, pr.dt)
    op = orc.OracleProblem(Q, R, QT, xr, ur, pr.dt)
    return pr, bp, op


def _one_iteration_vs_oracle(aoc, bp, op, x0, kk, prm_kw, oprm_kw, n_iter=1):
    B = x0.shape[0]
    prm = aoc.make_params(**prm_kw)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    s.set_initial_from_x0(x0)
    oprm = orc.params(**oprm_kw)
    xi, ui = s.current()
    for it in range(n_iter):
        s.iterate(kk + it)
        sc = s.scalars()
        xn, un = s.current()
        for b in range(B):
            r = orc.newton_iterate(op, oprm, kk + it, xi[b], ui[b], xi[b][:, 0])
            assert r["stepsize"] == sc["stepsize"][b] and r["ntrials"] == sc["ntrials"][b], (it, b, r["stepsize"], sc["stepsize"][b])
            assert abs(r["J"] - sc["cost"][b]) <= 1e-12 * abs(r["J"]) + 1e-300
            assert abs(r["descent"] - sc["descent"][b]) <= 1e-8 * abs(r["descent"]) + 1e-300
            assert np.array_equal(xn[b], r["xx"]), (it, b)
            assert rel_err(un[b], r["uu"], 1e-3) < 1e-8, (it, b)
        xi, ui = xn, un
    return s


@pytest.mark.parametrize("large_batch_kernels", [False, True])
@pytest.mark.parametrize("T", [3, 4, 5, 9, 17])
def test_tiny_horizons(aoc, tuned, T, large_batch_kernels):
    """large_batch_kernels: the one-wavefront-per-tile kernels (k_backward with its peeled stage 0 and prefetch ring,
    k_forward with two candidates riding along, k_ls_final) instead of the multi-wavefront ones a batch this small gets."""
    from aircraftoptimalcontrol_amd import problems
    if large_batch_kernels:
        tuned(split_tiles=0, split_bw_tiles=0, nspec=2, ls_worklist=1)
    pr, bp, op = _setup(aoc, T)
    x0 = problems.perturbed_x0(pr, 5, seed=T)
    kw = dict(stepsize_0=1.0, armijo_maxiters=10)
    _one_iteration_vs_oracle(aoc, bp, op, x0, 0, kw, {}, n_iter=2)
    _one_iteration_vs_oracle(aoc, bp, op, x0, 9, kw, {})   # full Hessian branch


@pytest.mark.parametrize("B", [1, 63, 64, 65, 129])
def test_ragged_batches(aoc, B):
    from aircraftoptimalcontrol_amd import problems
    pr, bp, op = _setup(aoc, 60)
    x0 = problems.random_x0(B, seed=5)
    s = aoc.NewtonBatchSolver(bp, B, aoc.make_params(stepsize_0=1.0, armijo_maxiters=10))
    s.set_initial_from_x0(x0)
    xi, ui = s.current()
    s.iterate(0)
    sc = s.scalars()
    assert all(v.shape[0] == B for v in sc.values())
    oprm = orc.params()
    for b in sorted({0, B // 2, B - 1}):
        r = orc.newton_iterate(op, oprm, 0, xi[b], ui[b], xi[b][:, 0])
        assert r["stepsize"] == sc["stepsize"][b] and abs(r["descent"] - sc["descent"][b]) <= 1e-8 * abs(r["descent"])


def test_dense_weights(aoc):
    """General dense Q, R, Q_T (the reference's Cost takes any matrices, aircraft_simplified.py:61)."""
    from aircraftoptimalcontrol_amd import problems
    pr, bp, op = _setup(aoc, 120, dense=True)
    x0 = problems.perturbed_x0(pr, 6, seed=1)
    kw = dict(stepsize_0=1.0, armijo_maxiters=10)
    _one_iteration_vs_oracle(aoc, bp, op, x0, 0, kw, {}, n_iter=3)
    _one_iteration_vs_oracle(aoc, bp, op, x0, 9, kw, {})


def test_reference_constructor_defaults(aoc):
    """NewtonMethod defaults: stepsize_0 = 1e-2, armijo_maxiters = 20 (optcon.py:335-339)."""
    from aircraftoptimalcontrol_amd import problems
    pr, bp, op = _setup(aoc, 100)
    x0 = problems.perturbed_x0(pr, 4, seed=2)
    _one_iteration_vs_oracle(aoc, bp, op, x0, 0, dict(), dict(stepsize_0=1e-2, armijo_maxiters=20), n_iter=2)


def test_line_search_exhaustion_and_small_maxiters(aoc):
    """armijo_maxiters = 1, 2, 3 (fewer than the trials riding along in the forward pass) and a cc so
    strict that every search is exhausted: the never-evaluated step is applied (Q5)."""
    from aircraftoptimalcontrol_amd import _lib, problems
    pr, bp, op = _setup(aoc, 80)
    x0 = problems.perturbed_x0(pr, 70, seed=4)
    for mi in (1, 2, 3, 5):
        kw = dict(stepsize_0=1.0, armijo_maxiters=mi, cc=0.5, beta=0.7)
        _one_iteration_vs_oracle(aoc, bp, op, x0[:4], 0, kw, dict(armijo_maxiters=mi))
    kw = dict(stepsize_0=1.0, armijo_maxiters=4, cc=1e6, beta=0.7)
    s = _one_iteration_vs_oracle(aoc, bp, op, x0, 0, kw, dict(armijo_maxiters=4, cc=1e6))
    sc = s.scalars()
    assert np.all(sc["stepsize"] == 0.7 * (0.7 * (0.7 * (0.7 * 1.0)))) and np.all(sc["ntrials"] == 4)
    assert np.all(sc["status"] & _lib.ST_ARMIJO_EXH)


def test_argument_errors(aoc):
    from aircraftoptimalcontrol_amd import problems
    pr, bp, op = _setup(aoc, 50)
    s = aoc.NewtonBatchSolver(bp, 3)
    with pytest.raises(ValueError):
        s.set_initial(np.zeros((3, 6, 49)), np.zeros((3, 2, 50)))
    with pytest.raises(ValueError):
        aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, np.zeros((5, 50)), np.zeros((2, 50)), 1e-3)
    # armijo_maxiters beyond the step table
    s2 = aoc.NewtonBatchSolver(bp, 3, aoc.make_params(armijo_maxiters=64))
    s2.set_initial_from_x0(problems.perturbed_x0(pr, 3))
    with pytest.raises(aoc.AocError):
        s2.iterate(0)


def test_nan_and_v_nonpositive_are_flagged_not_fatal(aoc):
    """V <= 0 divides by zero in the reference (aircraft_simplified.py:310) and NaNs then propagate
    silently; here the trajectory is flagged and its neighbours in the tile are untouched."""
    from aircraftoptimalcontrol_amd import _lib, problems
    pr, bp, op = _setup(aoc, 60)
    x0 = problems.perturbed_x0(pr, 64, seed=9)
    good = x0.copy()
    x0[7, 2] = 0.0          # V = 0
    x0[40, 2] = -3.0        # V < 0
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, 64, prm)
    s.set_initial_from_x0(x0)
    s.iterate(0)
    st = s.scalars()["status"]
    x_bad, u_bad = s.current()
    s.set_initial_from_x0(good)
    s.iterate(0)
    x_ok, u_ok = s.current()
    assert st[7] & (_lib.ST_NAN | _lib.ST_VNONPOS) and st[40] & (_lib.ST_NAN | _lib.ST_VNONPOS)
    keep = np.ones(64, bool); keep[[7, 40]] = False
    assert not st[keep].any()
    assert np.array_equal(x_bad[keep], x_ok[keep]) and np.array_equal(u_bad[keep], u_ok[keep])


@pytest.mark.parametrize("chunk_mb", [None, 64])
def test_solver_on_a_vmm_arena_equals_the_solver_on_torch_memory(aoc, chunk_mb):
    """vmm.Arena (HIP virtual-memory management: a reservation backed by physical handles of a stated size, handed out as
    torch tensors) as the home of a solver's big streams: the same results bit for bit — where memory lives decides speed
    (DESIGN.md section 4 "placement"), never values — and the arena's bookkeeping (one handle per stream / uniform chunks)."""
    import torch
    from aircraftoptimalcontrol_amd import problems, vmm
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 200
    x0 = problems.perturbed_x0(pr, B, seed=9)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    chunk = None if chunk_mb is None else chunk_mb << 20
    ar = vmm.Arena(bp.device, aoc.solver_arena_bytes(B, pr.T, chunk=chunk), chunk_bytes=chunk)
    res = []
    for arena in (None, ar):
        s = aoc.NewtonBatchSolver(bp, B, prm, arena=arena)
        s.set_initial_from_x0(x0)
        for kk in range(3):
            s.iterate(kk)
        res.append((s.current(), s.scalars()))
        if arena is not None:
            inside = lambda t: any(sg[0] <= t.data_ptr() and t.data_ptr() + t.numel() * t.element_size() <= sg[0] + sg[2] for sg in ar.segments)
            assert all(inside(t) for t in [s.ws] + s.xb + s.ub)
            assert len(ar.handles) == (7 if chunk is None else -(-ar.segments[0][2] // chunk))
        del s
    (xa, ua), sa = res[0]
    (xb, ub), sb = res[1]
    assert np.array_equal(xa, xb) and np.array_equal(ua, ub)
    for key in sa:
        assert np.array_equal(sa[key], sb[key]), key
    torch.cuda.synchronize()
    ar.close()
