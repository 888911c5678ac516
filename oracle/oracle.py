"""ctypes front-end of the CPU oracle (oracle/aoc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package never does.  Arrays use the reference's single-trajectory conventions:
xx (6,T), uu (2,T) float64 C-order; batched (B,6,T)/(B,2,T).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "build", "liboracle.so")


def build(force=False):
    src = os.path.join(_HERE, "aoc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


class Model(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("cd0", "cda", "cla", "m", "g", "S", "rho", "J", "dt")]


class Problem(C.Structure):
    _fields_ = [("mdl", Model), ("QQt", C.c_double * 36), ("RRt", C.c_double * 4), ("QQT", C.c_double * 36),
                ("T", C.c_int32), ("pad_", C.c_int32), ("xx_ref", C.c_void_p), ("uu_ref", C.c_void_p)]


class Params(C.Structure):
    _fields_ = [("max_iters", C.c_int32), ("armijo_maxiters", C.c_int32), ("stepsize_0", C.c_double),
                ("cc", C.c_double), ("beta", C.c_double), ("term_cond", C.c_double),
                ("hessian_switch", C.c_int32), ("pad_", C.c_int32)]


_lib = None
_portable = None
_SO_NATIVE = os.path.join(_HERE, "build", "liboracle_native.so")
NATIVE_FLAGS = "-O3 -march=native -fno-tree-slp-vectorize -ffp-contract=off -fno-fast-math -fopenmp"
PORTABLE_FLAGS = "-O2 -ffp-contract=off -fno-fast-math -fopenmp"
flags = PORTABLE_FLAGS          # of the library lib() returns


def _load(path):
    l = C.CDLL(path)
    l.orc_stagecost.restype = C.c_double
    l.orc_termcost.restype = C.c_double
    l.orc_traj_cost.restype = C.c_double
    l.orc_armijo.restype = C.c_double
    return l


def lib():
    global _lib, _portable
    if _lib is None:
        build()
        _lib = _portable = _load(_SO)
    return _lib


def use_native(on=True):
    """Switch lib() to the build for THIS host (`make native`: -O3 -march=native, same floating-point rules) — what
    bench.py's cpu_baseline times — or back to the portable -O2 build.  The native library is compiled here
    and now (it cannot travel: the build container's CPU is not the GPU box's) and is accepted only if one Newton
    iteration and a four-iteration chain on a few trajectories come out bit for bit as from the portable build.  Returns
    the flags of the library now in use; raises if the native build cannot be made or differs."""
    global _lib, flags
    lib()
    if not on:
        _lib, flags = _portable, PORTABLE_FLAGS
        return flags
    subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])
    nat = _load(_SO_NATIVE)
    # self-check on synthetic inputs (no fixture needed): a smooth reference, perturbed starts
    T, B = 120, 6
    tt = np.linspace(0.0, 1.0, T)
    xr = np.zeros((6, T)); xr[0] = 15.0 * tt; xr[1] = 1.0 / (1.0 + np.exp(-12 * (tt - 0.5))); xr[2] = 15.0
    ur = np.zeros((2, T)); ur[0] = 46.0
    Q = np.diag([1.0, 10.0, 0.1, 0.1, 0.1, 0.1]); R = np.diag([1e-3, 1e-2]); QT = 10 * Q
    prob = OracleProblem(Q, R, QT, xr, ur, 1.0 / T)
    prm = params()
    rng = np.random.default_rng(7)
    x0 = np.array([0.0, 0.0, 15.0, 0.05, 0.0, 0.02]) + rng.normal(0, 1, (B, 6)) * np.array([0.3, 0.3, 0.5, 0.02, 0.05, 0.02])
    res = []
    for l in (_portable, nat):
        _lib = l
        X, U = initial_guess_batch(prob.c.mdl, xr, x0, nthreads=2)
        h = newton_iterate_batch(prob, prm, X, U, X[:, :, 0].copy(), 7, 4, nthreads=2)   # kk = 7..10: both Hessian regimes
        res.append((X.copy(), U.copy(), h["cost"].copy(), h["descent"].copy(), h["stepsize"].copy(), h["ntrials"].copy()))
    same = all(np.array_equal(a, b, equal_nan=True) for a, b in zip(*res))
    if not same:
        _lib, flags = _portable, PORTABLE_FLAGS
        raise RuntimeError("the -march=native build of the oracle does not reproduce the portable build bit for bit")
    _lib, flags = nat, NATIVE_FLAGS
    return flags


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def default_model(dt=1e-3):
    m = Model()
    lib().orc_default_model(C.byref(m))
    m.dt = dt
    return m


class OracleProblem:
    """Holds the C problem struct and keeps the reference arrays alive."""

    def __init__(self, QQt, RRt, QQT, xx_ref, uu_ref, dt, model=None):
        self.xx_ref, self.uu_ref = _f64(xx_ref), _f64(uu_ref)
        self.T = self.xx_ref.shape[1]
        assert self.xx_ref.shape == (6, self.T) and self.uu_ref.shape == (2, self.T)
        self.c = Problem()
        self.c.mdl = model if model is not None else default_model(dt)
        self.c.mdl.dt = dt
        self.c.QQt[:] = _f64(QQt).ravel().tolist()
        self.c.RRt[:] = _f64(RRt).ravel().tolist()
        self.c.QQT[:] = _f64(QQT).ravel().tolist()
        self.c.T = self.T
        self.c.xx_ref = self.xx_ref.ctypes.data
        self.c.uu_ref = self.uu_ref.ctypes.data


def params(max_iters=200, stepsize_0=1.0, cc=0.5, beta=0.7, armijo_maxiters=10, term_cond=-1e-6,
           hessian_switch=8):
    return Params(max_iters, armijo_maxiters, stepsize_0, cc, beta, term_cond, hessian_switch, 0)


def step(model, x, u, lmbd=None, want3=False):
    """Dynamics.step.  Returns (xp[f32 as f64], fx(6,6)=A^T, fu(2,6)=B^T, fxx, fuu, fux)."""
    x, u = _f64(x), _f64(u)
    xp = np.zeros(6); fx = np.zeros((6, 6)); fu = np.zeros((2, 6))
    fxx = np.zeros((6, 6)); fuu = np.zeros((2, 2)); fux = np.zeros((2, 6))
    fxx3 = np.zeros((6, 6, 6)) if want3 else None
    fux3 = np.zeros((2, 6, 6)) if want3 else None
    lm = None if lmbd is None else _f64(lmbd)
    lib().orc_step(C.byref(model), _p(x), _p(u), _p(lm), _p(xp), _p(fx), _p(fu), _p(fxx), _p(fuu), _p(fux),
                   _p(fxx3), _p(fux3))
    if want3:
        return xp, fx, fu, fxx3, np.zeros((2, 2, 6)), fux3
    return xp, fx, fu, fxx, fuu, fux


def stagecost(prob, x, u, xr, ur):
    lx = np.zeros(6); lu = np.zeros(2)
    ll = lib().orc_stagecost(C.byref(prob.c), _p(_f64(x)), _p(_f64(u)), _p(_f64(xr)), _p(_f64(ur)), _p(lx), _p(lu))
    return ll, lx, lu


def termcost(prob, x, xr):
    lTx = np.zeros(6)
    ll = lib().orc_termcost(C.byref(prob.c), _p(_f64(x)), _p(_f64(xr)), _p(lTx))
    return ll, lTx


def traj_cost(prob, xx, uu):
    return lib().orc_traj_cost(C.byref(prob.c), _p(_f64(xx)), _p(_f64(uu)))


def get_update(prob, stepsize, uu, du, x0):
    T = prob.T
    xx_t = np.zeros((6, T)); uu_t = np.zeros((2, T))
    lib().orc_get_update(C.byref(prob.c), C.c_double(stepsize), _p(_f64(uu)), _p(_f64(du)), _p(_f64(x0)),
                         _p(xx_t), _p(uu_t))
    return xx_t, uu_t


def ltv_lqr(AA, BB, QQ, RR, SS, QQf, x0, qq=None, rr=None, qqf=None):
    """Inputs in the reference's layout (.., .., T); outputs in the reference's layout too."""
    T = AA.shape[2]
    tm = lambda a: _f64(np.moveaxis(a, -1, 0))
    aug = qq is not None or rr is not None or qqf is not None
    n = 7 if aug else 6
    KK = np.zeros((T, 2, n)); PP = np.zeros((T, n, n)); xo = np.zeros((T, 6)); uo = np.zeros((T, 2))
    nreg = C.c_int(0)
    a = [tm(AA), tm(BB), tm(QQ), tm(RR), tm(SS), _f64(QQf), _f64(x0)]
    q = None if qq is None else tm(qq)
    r = None if rr is None else tm(rr)
    qf = None if qqf is None else _f64(qqf)
    ns = lib().orc_ltv_lqr(T, *[_p(v) for v in a], _p(q), _p(r), _p(qf), _p(KK), _p(PP), _p(xo), _p(uo),
                           C.byref(nreg))
    return (np.moveaxis(KK, 0, -1), np.moveaxis(PP, 0, -1), xo.T.copy(), uo.T.copy(), nreg.value, ns)


def newton_iterate(prob, prm, kk, xx, uu, x0, want_internals=False):
    T = prob.T
    xn = np.zeros((6, T)); un = np.zeros((2, T))
    J = C.c_double(); d = C.c_double(); s = C.c_double(); ntr = C.c_int(); nreg = C.c_int()
    KK = np.zeros((T, 2, 7)) if want_internals else None
    du = np.zeros((2, T)) if want_internals else None
    dx = np.zeros((6, T)) if want_internals else None
    lm = np.zeros((6, T)) if want_internals else None
    ns = lib().orc_newton_iterate(C.byref(prob.c), C.byref(prm), int(kk), _p(_f64(xx)), _p(_f64(uu)), _p(_f64(x0)),
                                  _p(xn), _p(un), C.byref(J), C.byref(d), C.byref(s), C.byref(ntr), C.byref(nreg),
                                  _p(KK), _p(du), _p(dx), _p(lm))
    out = dict(xx=xn, uu=un, J=J.value, descent=d.value, stepsize=s.value, ntrials=ntr.value,
               nreg=nreg.value, nsing=ns)
    if want_internals:
        out.update(KK=KK, du=du, dx=dx, lmbd=lm)
    return out


def gradient_iterate(prob, prm, xx, uu, x0):
    """One iteration of GradientMethod.optimize with the missing JP argument supplied (parity unpinned: the
    reference's method raises TypeError)."""
    T = prob.T
    xn = np.zeros((6, T)); un = np.zeros((2, T)); du = np.zeros((2, T))
    J = C.c_double(); d = C.c_double(); s = C.c_double(); ntr = C.c_int()
    lib().orc_gradient_iterate(C.byref(prob.c), C.byref(prm), _p(_f64(xx)), _p(_f64(uu)), _p(_f64(x0)), _p(xn), _p(un),
                               C.byref(J), C.byref(d), C.byref(s), C.byref(ntr), _p(du))
    return dict(xx=xn, uu=un, J=J.value, descent=d.value, stepsize=s.value, ntrials=ntr.value, du=du)


def newton_optimize(prob, prm, xx_init, uu_init):
    T = prob.T
    xs = np.zeros((6, T)); us = np.zeros((2, T))
    n = prm.max_iters
    hJ = np.zeros(n); hd = np.zeros(n); hs = np.zeros(n); hn = np.zeros(n, np.int32)
    it = C.c_int32(); ri = C.c_int32()
    ns = lib().orc_newton_optimize(C.byref(prob.c), C.byref(prm), _p(_f64(xx_init)), _p(_f64(uu_init)), _p(xs), _p(us),
                                   _p(hJ), _p(hd), _p(hs), _p(hn), C.byref(it), C.byref(ri))
    k = it.value
    return dict(xx_star=xs, uu_star=us, cost=hJ[:k], descent=hd[:k], stepsize=hs[:k], ntrials=hn[:k],
                iters=k, ret_index=ri.value, nsing=ns)


def initial_trajectory(model, xx_ref):
    xr = _f64(xx_ref); T = xr.shape[1]
    xx = np.zeros((6, T)); uu = np.zeros((2, T))
    lib().orc_initial_trajectory(C.byref(model), T, _p(xr), _p(xx), _p(uu))
    return xx, uu


def initial_guess_batch(model, xx_ref, x0, nthreads=0):
    """P-controller initial guesses from every x0[b] (B,6) against the shared xx_ref (6,T)."""
    xr, x0 = _f64(xx_ref), _f64(x0)
    B, T = x0.shape[0], xr.shape[1]
    xx = np.zeros((B, 6, T)); uu = np.zeros((B, 2, T))
    if nthreads <= 0:
        nthreads = lib().orc_max_threads()
    lib().orc_initial_trajectory_batch(C.byref(model), B, T, _p(xr), _p(x0), _p(xx), _p(uu), int(nthreads))
    return xx, uu


def lqr_tracking(model, QQt, RRt, QQT, xx_opt, uu_opt, delta):
    xo, uo = _f64(xx_opt), _f64(uu_opt); T = xo.shape[1]
    xr = np.zeros((6, T)); ur = np.zeros((2, T)); KK = np.zeros((T, 2, 6))
    ns = lib().orc_lqr_tracking(C.byref(model), T, _p(_f64(QQt)), _p(_f64(RRt)), _p(_f64(QQT)), _p(xo), _p(uo),
                                _p(_f64(delta)), _p(xr), _p(ur), _p(KK))
    return xr, ur, np.moveaxis(KK, 0, -1), ns


def newton_iterate_batch(prob, prm, xx, uu, x0, kk0, n_iters, nthreads=0):
    """In-place over xx (B,6,T), uu (B,2,T).  Returns per-iteration scalars (B,n_iters)."""
    assert xx.flags.c_contiguous and uu.flags.c_contiguous and xx.dtype == np.float64
    B = xx.shape[0]
    J = np.zeros((B, n_iters)); d = np.zeros((B, n_iters)); s = np.zeros((B, n_iters))
    n = np.zeros((B, n_iters), np.int32)
    nreg = np.zeros((B, n_iters), np.int32)     # stages whose M was regularised in that iteration (optcon.py:745-749)
    if nthreads <= 0:
        nthreads = lib().orc_max_threads()
    ns = lib().orc_newton_iterate_batch(C.byref(prob.c), C.byref(prm), B, int(kk0), int(n_iters), _p(xx), _p(uu),
                                        _p(_f64(x0)), _p(J), _p(d), _p(s), _p(n), _p(nreg), int(nthreads))
    return dict(cost=J, descent=d, stepsize=s, ntrials=n, nreg=nreg, nsing=ns, nthreads=nthreads)
