#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

This script only works in the build container, where the upstream repository is
mounted read-only at /root/reference.  It imports the reference's three library
modules (optcon, aircraft_simplified, lqr_tracking) headlessly, drives them with
the parameters of the reference's own driver scripts, and saves inputs + outputs
as small .npz fixtures.  Nothing of the reference travels: fixtures hold numbers
only.  The driver scripts themselves are NOT imported (they need cvxpy and call
plt.show/np.save at import); the problem set-up they do is re-stated below from
main_newton_method.py:32-142 and acrobatic_newton.py:34-154.

Run:  MPLBACKEND=Agg PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_golden.py [config1]
"""
import contextlib
import io
import os
import re
import sys
import warnings

import numpy as np

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)
warnings.simplefilter("ignore")

import aircraft_simplified as ref_air  # noqa: E402
import lqr_tracking as ref_lqr  # noqa: E402
import optcon as ref_opt  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------------------
# problem set-up (parameters of the reference's drivers)
# --------------------------------------------------------------------------------------
def sigmoid(tt, slope):
    ss = 1 / (1 + np.exp((-tt) * slope))
    return ss, ss * (1 - ss)


def step_problem(tf=1.0, dt=1e-3):
    """main_newton_method.py:50-142 (weights, sigmoid reference, equilibrium input)."""
    dyn = ref_air.Dynamics()
    ns, ni = dyn.ns, dyn.ni
    QQt = np.eye(ns) * 1e-6
    QQt[1, 1] = dyn.m * dyn.g * 0.01
    QQt[2, 2] = 0.5 * dyn.m * 0.001
    QQt[3, 3] = 0.01
    QQt[4, 4] = 0.5 * dyn.J * 0.001
    RRt = 1e-6 * np.eye(ni)
    QQT = QQt.copy()
    QQT[1, 1] = QQT[1, 1] * 20
    QQT[3, 3] = QQT[1, 1]
    QQT[0, 0] = QQT[1, 1]
    dyn.dt = dt
    TT = int(tf / dt)
    tt = np.linspace(0, tf, TT)
    x0, z0, xf, zf = 0, 0, 16, 2.71
    slope = tt.shape[0] * 1
    zz = z0 + sigmoid(tt - tt[-1] / 2, slope)[0] * (zf - z0)
    zzd = sigmoid(tt - tt[-1] / 2, slope)[1] * (zf - z0)
    xxe, uue = dyn.get_equilibrium(np.zeros(ns,), tt)
    xx_ref = np.zeros((ns, TT))
    uu_ref = np.zeros((ni, TT))
    xx_ref[0, :] = x0 + ((xf - x0) / tf) * tt
    xx_ref[1, :] = zz
    xx_ref[2, :] = (zzd ** 2 + ((xf - x0) / tf) ** 2) ** 0.5
    for i in range(ni):
        uu_ref[i, :] = uue[i]
    return dict(dyn=dyn, QQt=QQt, RRt=RRt, QQT=QQT, xx_ref=xx_ref, uu_ref=uu_ref,
                tt=tt, tf=tf, dt=dt, TT=TT, xxe=np.asarray(xxe, float), uue=np.asarray(uue))


def acro_problem(tf=1.0, dt=1e-3):
    """acrobatic_newton.py:52-154."""
    dyn = ref_air.Dynamics()
    ns, ni = dyn.ns, dyn.ni
    QQt = np.eye(ns) * 1e-6
    QQt[1, 1] = dyn.m * dyn.g * 0.01
    QQt[2, 2] = 0.5 * dyn.m * 0.001
    QQt[3, 3] = 0.01
    QQt[4, 4] = 0.5 * dyn.J * 0.001
    RRt = 1e-6 * np.eye(ni)
    QQT = QQt.copy()
    QQT[1, 1] = QQT[1, 1] * 100
    QQT[3, 3] = QQT[1, 1]
    QQT[0, 0] = QQT[1, 1]
    dyn.dt = dt
    TT = int(tf / dt)
    tt = np.linspace(0, tf, TT)
    x0, z0, xf, zf = 0, 0, 18, 2.71
    slope = tt.shape[0] * 0.1
    pp = np.zeros((TT,))
    pp[:TT // 2] = z0 + sigmoid(tt[:TT // 2] - tt[TT // 2] / 2, slope)[0] * (zf - z0)
    pp[TT // 2:] = z0 + sigmoid(-tt[:TT // 2] + tt[TT // 2] / 2, slope)[0] * (zf - z0)
    temp = pp.copy()
    N = pp.shape[0]
    pp = np.zeros(temp.shape)
    pp[int(0.05 * N):int(0.50 * N)] = temp[:int(0.45 * N)]
    pp[int(0.50 * N):int(0.95 * N)] = temp[-int(0.45 * N):]
    xxe, uue = dyn.get_equilibrium(np.zeros(ns,), tt)
    xx_ref = np.zeros((ns, TT))
    uu_ref = np.zeros((ni, TT))
    xx_ref[0, :] = x0 + ((xf - x0) / tf) * tt
    xx_ref[1, :] = pp
    for i in range(2, ns):
        xx_ref[i, :] = xxe[i]
    xx_ref[3, :] = 0.0
    for i in range(ni):
        uu_ref[i, :] = uue[i]
    uu_ref[0, :] = uu_ref[0, :] * 10
    uu_ref[1, :] = -60
    return dict(dyn=dyn, QQt=QQt, RRt=RRt, QQT=QQT, xx_ref=xx_ref, uu_ref=uu_ref,
                tt=tt, tf=tf, dt=dt, TT=TT, xxe=np.asarray(xxe, float), uue=np.asarray(uue))


def problem_arrays(p):
    return dict(QQt=p["QQt"], RRt=p["RRt"], QQT=p["QQT"], xx_ref=p["xx_ref"], uu_ref=p["uu_ref"],
                dt=np.float64(p["dt"]), tf=np.float64(p["tf"]), TT=np.int64(p["TT"]))


# --------------------------------------------------------------------------------------
# running the reference optimiser with taps on its own methods
# --------------------------------------------------------------------------------------
ITER_RE = re.compile(r"Iter = (\d+)\t Descent = (\S+)\t Cost = (\S+)")


def run_newton(p, xx_init, uu_init, n_iters, keep, armijo_maxiters=10, stepsize_0=1, cc=0.5, beta=0.7):
    """Run NewtonMethod(max_iters=n_iters+1).optimize; record every iterate the reference
    produces (return values of its get_update), the Armijo step sizes and the stdout scalars."""
    cst = ref_air.Cost(p["QQt"], p["RRt"], p["QQT"])
    NM = ref_opt.NewtonMethod(p["dyn"], cst, p["xx_ref"], p["uu_ref"], max_iters=n_iters + 1,
                              stepsize_0=stepsize_0, cc=cc, beta=beta,
                              armijo_maxiters=armijo_maxiters, term_cond=1e-6, visu_armijo=False)
    iterates, steps, ntrials = [], [], []
    orig_update, orig_armijo = NM.get_update, NM.armijo_stepsize
    ncalls = {"n": 0}
    dyn = p["dyn"]
    orig_step = dyn.step

    def tap_step(*a):
        ncalls["n"] += 1
        return orig_step(*a)

    def tap_update(stepsize, uu, deltau, x0):
        xx_t, uu_t = orig_update(stepsize, uu, deltau, x0)
        iterates.append((xx_t.copy(), uu_t.copy()))
        return xx_t, uu_t

    def tap_armijo(*a):
        n0 = ncalls["n"]
        s = orig_armijo(*a)
        ntrials.append((ncalls["n"] - n0) // (p["TT"] - 1))
        steps.append(float(s))
        return s

    NM.get_update, NM.armijo_stepsize, dyn.step = tap_update, tap_armijo, tap_step
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            xx_star, uu_star = NM.optimize(xx_init.copy(), uu_init.copy(), p["tf"], p["dt"])
    finally:
        dyn.step = orig_step
    xx_star, uu_star = xx_star.copy(), uu_star.copy()
    txt = buf.getvalue()
    sc = [(int(a), float(b), float(c)) for a, b, c in ITER_RE.findall(txt)]
    out = dict(descent=np.array([s[1] for s in sc]), cost=np.array([s[2] for s in sc]),
               stepsize=np.array(steps), ntrials=np.array(ntrials, dtype=np.int64),
               xx_star=xx_star, uu_star=uu_star, n_done=np.int64(len(sc)),
               xx_init=xx_init, uu_init=uu_init)
    for k in keep:
        if 1 <= k <= len(iterates):
            out["xx_it%d" % k] = iterates[k - 1][0]
            out["uu_it%d" % k] = iterates[k - 1][1]
    return out, iterates


def backward_quantities(p, xx, uu, full_hessian):
    """The linearisation loop of optcon.py:429-464, executed with the reference's own functions."""
    dyn = p["dyn"]
    cst = ref_air.Cost(p["QQt"], p["RRt"], p["QQT"])
    ns, ni, TT = 6, 2, p["TT"]
    xr, ur = p["xx_ref"], p["uu_ref"]
    QQ = np.zeros((ns, ns, TT)); RR = np.zeros((ni, ni, TT)); SS = np.zeros((ni, ns, TT))
    qq = np.zeros((ns, TT)); rr = np.zeros((ni, TT))
    AA = np.zeros((ns, ns, TT)); BB = np.zeros((ns, ni, TT)); lmbd = np.zeros((ns, TT))
    lT, QT = cst.termcost(xx[:, TT - 1], xr[:, TT - 1])[1:]
    lmbd[:, TT - 1] = lT.squeeze(); QQ[:, :, TT - 1] = QT; qq[:, TT - 1] = lT.squeeze()
    for t in reversed(range(TT - 1)):
        aa, bb, lxx, lxu, lux, luu = cst.stagecost(xx[:, t], uu[:, t], xr[:, t], ur[:, t])[1:]
        fx, fu, fxx, fuu, fux = dyn.step(xx[:, t], uu[:, t], lmbd[:, t + 1])[1:]
        A, Bm = fx.T, fu.T
        if full_hessian:
            QQ[:, :, t] = lxx + fxx; RR[:, :, t] = luu + fuu; SS[:, :, t] = lux + fux
        else:
            QQ[:, :, t] = lxx; RR[:, :, t] = luu; SS[:, :, t] = lux
        AA[:, :, t] = A; BB[:, :, t] = Bm; qq[:, t] = aa.squeeze(); rr[:, t] = bb.squeeze()
        lmbd[:, t] = (A.T @ lmbd[:, t + 1][:, None] + aa).squeeze()
    return AA, BB, QQ, RR, SS, qq, rr, lmbd


def lqr_case(p, xx, uu, full_hessian):
    AA, BB, QQ, RR, SS, qq, rr, lmbd = backward_quantities(p, xx, uu, full_hessian)
    TT = p["TT"]
    nreg = {"n": 0}
    orig = np.linalg.eigvals

    def tap(M):
        w = orig(M)
        if not np.all(w > 0):
            nreg["n"] += 1
        return w

    np.linalg.eigvals = tap
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            KK, PP, dx, du = ref_opt.ltv_LQR(AA, BB, QQ, RR, SS, QQ[:, :, TT - 1], TT, np.zeros(6),
                                             qq, rr, qq[:, TT - 1])
    finally:
        np.linalg.eigvals = orig
    descent = 0.0
    for t in reversed(range(TT - 1)):
        descent += float(((BB[:, :, t].T @ lmbd[:, t + 1][:, None] + rr[:, t][:, None]).T
                          @ du[:, t].reshape((2, 1))).item())
    return dict(xx=xx, uu=uu, full_hessian=np.int64(full_hessian), KK=KK, PP_s=PP[:, :, ::50].copy(),
                dx=dx.copy(), du=du.copy(), lmbd=lmbd, descent=np.float64(descent),
                n_regularised=np.int64(nreg["n"]))


def full_solve(p, name):
    """G8: NewtonMethod(max_iters=200).optimize to its own termination — iteration count, Armijo steps and trial
    counts, stdout scalars, the returned iterate (index n_done-2, Q7) and its neighbours."""
    xi, ui = p["dyn"].get_initial_trajectory(p["xx_ref"], p["tt"])
    full, its = run_newton(p, xi, ui, 199, ())
    nd = int(full["n_done"])
    for k in (nd - 2, nd - 1, nd):
        if 1 <= k <= len(its):
            full["xx_it%d" % k], full["uu_it%d" % k] = its[k - 1]
    save(name, **full)
    print("full solve %s: iterations printed = %d, cost %r -> %r, ntrials %s" %
          (name, nd, float(full["cost"][0]), float(full["cost"][-1]), full["ntrials"].tolist()))


def config1_full_solves():
    """BASELINE configs[0] / SURVEY 8d config 1: the two drivers of the reference at their native T = 1000
    (main_newton_method.py:32-142: 23 iterations, returned iterate 21; acrobatic_newton.py:34-154: 37 iterations)."""
    full_solve(step_problem(), "g8_full_step_T1000")
    full_solve(acro_problem(), "g8_full_acro_T1000")


def airfoil_vectors():
    """the post-processing module animate.py (:6-140).  The outline of Airfoil(20, ...) (the thickness both
    drivers use, main_newton_method.py:226, acrobatic_newton.py:242) and of a 12 % section, the outline at seeded random
    poses through update_pose, and what animate(i) puts into its artists for a few frames of a seeded trajectory (the
    artists are made here, as run_animation makes them, without rendering 1000 frames)."""
    import matplotlib.pyplot as plt
    import animate as ref_anim
    rng = np.random.default_rng(20261005)
    dt, TT = 0.05, 20
    xs = np.stack([np.linspace(0, 16, TT), 3 * np.sin(np.linspace(0, 3, TT)), rng.uniform(9, 23, TT),
                   rng.uniform(-3.2, 3.2, TT), rng.normal(0, 5, TT), rng.uniform(-1, 1, TT)])
    xr = xs + rng.normal(0, 0.3, xs.shape)
    af = ref_anim.Airfoil(20, xs, xr, dt=dt, xlim=[0, 17], ylim=[-5, 5])
    outline20 = af.update_pose(0.0, 0.0, 0.0)            # T = identity
    outline12 = ref_anim.Airfoil(12, xs, xr).update_pose(0.0, 0.0, 0.0)
    n = 16
    th, px, py = rng.uniform(-3.2, 3.2, n), rng.uniform(-2, 20, n), rng.uniform(-5, 5, n)
    posed = np.stack([af.update_pose(th[i], px[i], py[i]) for i in range(n)])
    fig, ax = plt.subplots()
    af.line0, = ax.plot([], []); af.line1, = ax.plot([], []); af.point1, = ax.plot([], [])
    af.time_template = 't = %.1f s'; af.time_text = ax.text(0, 0, '')
    frames = np.array([0, 7, 19])
    l0 = []; l1 = []
    point_error = ""
    for i in frames:
        try:
            af.animate(int(i))
        except RuntimeError as e:        # Matplotlib >= 3.7 refuses the scalars animate.py:111 hands to point1.set_data:
            point_error = str(e)         # the two outlines are set by then, the marker and the time label are not
        l0.append(np.stack(af.line0.get_data())); l1.append(np.stack(af.line1.get_data()))
    plt.close(fig)
    save("g10_airfoil", xx_star=xs, xx_ref=xr, dt=np.float64(dt), outline20=outline20, outline12=outline12,
         theta=th, x_loc=px, y_loc=py, posed=posed, frames=frames, line0=np.stack(l0), line1=np.stack(l1),
         point_error=np.array(point_error))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "airfoil":     # only the fixture added in round 5
        return airfoil_vectors()
    if len(sys.argv) > 1 and sys.argv[1] == "config1":   # only the fixtures added in round 2
        return config1_full_solves()
    rng = np.random.default_rng(20260331)

    # ---------------- G1: Dynamics.step unit vectors ----------------
    dyn = ref_air.Dynamics()
    n = 64
    X = np.stack([rng.uniform(-5, 20, n), rng.uniform(-3, 5, n), rng.uniform(9, 23, n),
                  rng.uniform(-1.5, 1.5, n), rng.uniform(-25, 25, n), rng.uniform(-1.5, 1.5, n)], 1)
    U = np.stack([rng.uniform(0, 900, n), rng.uniform(-100, 300, n)], 1)
    L = rng.normal(0, 50, (n, 6))
    for dt in (1e-3, 2e-3):
        dyn.dt = dt
        xp = np.zeros((n, 6), np.float32); fx = np.zeros((n, 6, 6)); fu = np.zeros((n, 2, 6))
        fxx = np.zeros((n, 6, 6)); fuu = np.zeros((n, 2, 2)); fux = np.zeros((n, 2, 6))
        fxx3 = np.zeros((n, 6, 6, 6)); fux3 = np.zeros((n, 2, 6, 6)); fuu3 = np.zeros((n, 2, 2, 6))
        for i in range(n):
            xp[i], fx[i], fu[i], fxx[i], fuu[i], fux[i] = dyn.step(X[i], U[i], L[i])
            r = dyn.step(X[i], U[i])
            assert np.array_equal(r[0], xp[i])
            fxx3[i], fuu3[i], fux3[i] = r[3], r[4], r[5]
        save("g1_step_dt%g" % dt, x=X, u=U, lmbd=L, dt=np.float64(dt), xp=xp, fx=fx, fu=fu,
             fxx=fxx, fuu=fuu, fux=fux, fxx3=fxx3[:8], fuu3=fuu3[:8], fux3=fux3[:8])

    # ---------------- G2: Cost unit vectors ----------------
    ps = step_problem()
    pa = acro_problem()
    for tag, p in (("step", ps), ("acro", pa)):
        cst = ref_air.Cost(p["QQt"], p["RRt"], p["QQT"])
        XR = X + rng.normal(0, 1, X.shape)
        UR = U + rng.normal(0, 10, U.shape)
        ll = np.zeros(n); lx = np.zeros((n, 6)); lu = np.zeros((n, 2)); lT = np.zeros(n); lTx = np.zeros((n, 6))
        for i in range(n):
            r = cst.stagecost(X[i], U[i], XR[i], UR[i])
            ll[i], lx[i], lu[i] = r[0].item(), r[1].squeeze(), r[2].squeeze()
            assert np.array_equal(r[3], p["QQt"]) and np.array_equal(r[6], p["RRt"])
            assert not r[4].any() and not r[5].any()
            rT = cst.termcost(X[i], XR[i])
            lT[i], lTx[i] = rT[0].item(), rT[1].squeeze()
        save("g2_cost_" + tag, x=X, u=U, xr=XR, ur=UR, QQt=p["QQt"], RRt=p["RRt"], QQT=p["QQT"],
             ll=ll, lx=lx, lu=lu, llT=lT, lTx=lTx)

    # ---------------- problems + reference initial guesses ----------------
    save("problem_step_T1000", xxe=ps["xxe"], uue=ps["uue"].astype(np.float64), **problem_arrays(ps))
    save("problem_acro_T1000", xxe=pa["xxe"], uue=pa["uue"].astype(np.float64), **problem_arrays(pa))
    p5 = step_problem(tf=1.0, dt=2e-3)
    save("problem_step_T500", xxe=p5["xxe"], uue=p5["uue"].astype(np.float64), **problem_arrays(p5))

    xi_s, ui_s = ps["dyn"].get_initial_trajectory(ps["xx_ref"], ps["tt"])
    xi_a, ui_a = pa["dyn"].get_initial_trajectory(pa["xx_ref"], pa["tt"])
    xi_5, ui_5 = p5["dyn"].get_initial_trajectory(p5["xx_ref"], p5["tt"])

    # ---------------- G6/G5: Newton chains with per-iteration taps ----------------
    keep = (1, 2, 3, 6, 9, 10, 11, 12, 16, 20)
    c5, its5 = run_newton(p5, xi_5, ui_5, 20, keep)
    save("g6_chain_step_T500", **c5)
    cs, itss = run_newton(ps, xi_s, ui_s, 12, (1, 3, 10, 12))
    save("g6_chain_step_T1000", **cs)
    ca, itsa = run_newton(pa, xi_a, ui_a, 10, (1, 10))
    save("g6_chain_acro_T1000", **ca)

    # ---------------- G8: full solves (termination + returned index quirk) ----------------
    full5, itf5 = run_newton(p5, xi_5, ui_5, 199, ())
    nd = int(full5["n_done"])
    # the reference returns iterate n_done-2 (Q7); keep neighbours so the test can tell them apart
    for k in (nd - 2, nd - 1, nd):
        if 1 <= k <= len(itf5):
            full5["xx_it%d" % k], full5["uu_it%d" % k] = itf5[k - 1]
    save("g8_full_step_T500", **full5)
    print("full T500 solve: iterations printed =", nd, "ntrials", full5["ntrials"].tolist())

    # ---------------- G3: ltv_LQR cases (a) GN on initial guess (b) full Hessian on initial guess
    #                  (regularisation fires) (c) full Hessian near the optimum ----------------
    for tag, p, xx, uu, fh in (("a_gn_init_T1000", ps, xi_s, ui_s, 0),
                               ("b_full_init_T1000", ps, xi_s, ui_s, 1),
                               ("c_full_near_T500", p5, its5[15][0], its5[15][1], 1),
                               ("d_gn_init_T500", p5, xi_5, ui_5, 0)):
        r = lqr_case(p, xx, uu, fh)
        print("  G3", tag, "regularised stages:", int(r["n_regularised"]), "descent", float(r["descent"]))
        save("g3_lqr_" + tag, **r)

    # ---------------- G4: lqr_tracking on the reference's saved optimum ----------------
    xx_opt = np.load(os.path.join(REF, "Data", "xx_star.npy"))
    uu_opt = np.load(os.path.join(REF, "Data", "uu_star.npy"))
    dynl = ref_air.Dynamics()
    QQl = np.eye(6) * 0.01; QQl[1, 1] = 10; QQl[0, 0] = 10
    RRl = np.eye(2) * 1e-5
    for k, v in dict(dyn=dynl, ns=6, ni=2, QQt=QQl, RRt=RRl, QQT=QQl.copy()).items():
        setattr(ref_lqr, k, v)
    ttl = np.linspace(0, 1, xx_opt.shape[1])
    xx_reg, uu_reg = ref_lqr.lqr_tracking(xx_opt, uu_opt, ttl)
    TT = xx_opt.shape[1]
    AA = np.zeros((6, 6, TT)); BB = np.zeros((6, 2, TT))
    for t in range(TT):
        _, fx, fu = dynl.step(xx_opt[:, t], uu_opt[:, t])[0:3]
        AA[:, :, t] = fx.T; BB[:, :, t] = fu.T
    KK, PP, lx, lu = ref_lqr.ltv_LQR(AA, BB, QQl, RRl, np.zeros((2, 6, TT)), QQl.copy(), TT,
                                     np.ones(6) * 0.1, None, None, None)
    KK2 = ref_opt.ltv_LQR(AA, BB, QQl, RRl, np.zeros((2, 6, TT)), QQl.copy(), TT,
                          np.ones(6) * 0.1, None, None, None)[0]
    assert np.array_equal(KK, KK2)
    save("g4_lqr_tracking", xx_opt=xx_opt, uu_opt=uu_opt, QQt=QQl, RRt=RRl, QQT=QQl, dt=np.float64(dynl.dt),
         xx_reg=xx_reg, uu_reg=uu_reg, KK=KK, PP_s=PP[:, :, ::50].copy(), lin_xx=lx, lin_uu=lu)
    # acrobatic optimum too (input fixture for config 3/5)
    save("data_acrobatic_star", xx_star=np.load(os.path.join(REF, "Data", "xx_star_acrobatic.npy")),
         uu_star=np.load(os.path.join(REF, "Data", "uu_star_acrobatic.npy")))

    # ---------------- G9: perturbed-x0 mini-batch, config-2 generation rule ----------------
    Bm = 8
    rng2 = np.random.default_rng(20260401)
    sig = np.array([0.5, 0.5, 0.5, 0.05, 0.1, 0.05])
    nit = 4
    x0s = p5["xx_ref"][:, 0][None, :] + rng2.normal(0, 1, (Bm, 6)) * sig
    XI = np.zeros((Bm, 6, p5["TT"])); UI = np.zeros((Bm, 2, p5["TT"]))
    XO = np.zeros((Bm, 6, p5["TT"])); UO = np.zeros((Bm, 2, p5["TT"]))
    CO = np.zeros((Bm, nit)); DE = np.zeros((Bm, nit)); ST = np.zeros((Bm, nit))
    for b in range(Bm):
        xr = p5["xx_ref"].copy(); xr[:, 0] = x0s[b]
        XI[b], UI[b] = p5["dyn"].get_initial_trajectory(xr, p5["tt"])
        r, its = run_newton(p5, XI[b], UI[b], nit, ())
        XO[b], UO[b] = its[nit - 1]
        CO[b], DE[b], ST[b] = r["cost"], r["descent"], r["stepsize"]
    config1_full_solves()
    airfoil_vectors()
    save("g9_minibatch_step_T500", x0=x0s, xx_init=XI, uu_init=UI, xx_out=XO.astype(np.float32),
         x0_out=XO[:, :, 0].copy(), uu_out=UO, cost=CO, descent=DE, stepsize=ST, n_iters=np.int64(nit))


if __name__ == "__main__":
    main()
