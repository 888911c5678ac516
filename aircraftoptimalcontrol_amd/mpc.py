"""Receding-horizon loop over the batched Newton solver (BASELINE.json configs[4], SURVEY 8d "Config 5").

The reference has no MPC; this is the loop SURVEY 8d defines on top of its pieces, every numerical
step being one of the reference's functions on the HIP library:

  per re-solve s, for every instance b (independent):
    1. gains     K = lqr_tracking-style LQR gains about the current optimum (lqr_tracking.py:268-276,
                 weights lqr_tracking.py:324-328)                                    -> aoc_lqr_tracking
    2. plant     u_cl = u_opt[0] + K[0] (x_true - x_opt[0])  (lqr_tracking.py:280);
                 x_true <- Dynamics.step(x_true, u_cl)[0] + disturbance             -> aoc_step_batch
    3. shift     u <- [u[1], ..., u[T-2], u[T-2], 0]; the reference window moves one sample
    4. warm start x0 <- x_true; initial iterate = rollout of the shifted u (get_update with
                 stepsize 0, optcon.py:176-200)                                       -> aoc_rollout_cost
    5. re-solve  n_newton iterations of NewtonMethod.optimize with kk restarting at 0
                 (Gauss-Newton Hessian only, optcon.py:443)                           -> aoc_newton_iterate

Host code here only moves pointers, shifts windows and draws the seeded disturbance.
"""
import ctypes as C

import numpy as np

from . import batch as _b
from ._lib import check, lib

TILE = _b.TILE


def window(ref_long, s, T):
    """Reference window [s, s+T) of a (C, >=T) curve, extended by holding its last sample."""
    n = ref_long.shape[1]
    idx = np.minimum(np.arange(s, s + T), n - 1)
    return ref_long[:, idx]


class RecedingHorizon:
    def __init__(self, pr, track_weights, B, T, n_newton=2, sigma=None, seed=20260405, device="cuda:0",
                 stepsize_0=1.0, cc=0.5, beta=0.7, armijo_maxiters=10):
        """pr: problems.ProblemData (weights + long reference curves xx_ref (6,L), uu_ref (2,L), L >= T);
        track_weights: (QQt, RRt, QQT) of the tracking LQR; sigma: (6,) std of the additive state
        disturbance per closed-loop step (None = none)."""
        torch = _b._torch()
        self.pr, self.B, self.T, self.n_newton = pr, int(B), int(T), int(n_newton)
        self.device = torch.device(device)
        self.tw = track_weights
        self.sigma = None if sigma is None else np.asarray(sigma, dtype=np.float64)
        self.seed = seed
        self.prm = _b.make_params(max_iters=n_newton + 1, stepsize_0=stepsize_0, cc=cc, beta=beta,
                                  armijo_maxiters=armijo_maxiters)
        self.s = 0
        self._set_window(0)
        self.solver = _b.NewtonBatchSolver(self.prob, self.B, self.prm)
        self.x_true = None
        self.Kg = _b.alloc_tiled(self.B, self.T, 12, self.device)

    def _set_window(self, s):
        pr, T = self.pr, self.T
        xr, ur = window(pr.xx_ref, s, T), window(pr.uu_ref, s, T)
        self.prob = _b.BatchProblem(pr.QQt, pr.RRt, pr.QQT, xr, ur, pr.dt, device=self.device)
        Q, R, QT = self.tw
        self.tprob = _b.BatchProblem(Q, R, QT, xr, ur, pr.dt, device=self.device)
        if hasattr(self, "solver"):
            self.solver.problem = self.prob

    def disturbance(self, s):
        """(B,6) additive state disturbance of closed-loop step s, keyed by (seed, s)."""
        if self.sigma is None:
            return np.zeros((self.B, 6))
        rng = np.random.Generator(np.random.Philox(key=self.seed, counter=[0, 0, 0, s]))
        return rng.normal(0, 1, (self.B, 6)) * self.sigma

    def start(self, x0, cold_iters=10):
        """Cold start: P-controller initial guess from x0 (B,6) and `cold_iters` Newton iterations."""
        self.solver.params = _b.make_params(max_iters=cold_iters + 1, stepsize_0=self.prm.stepsize_0, cc=self.prm.cc,
                                            beta=self.prm.beta, armijo_maxiters=self.prm.armijo_maxiters)
        self.solver.set_initial_from_x0(x0)
        for kk in range(cold_iters):
            self.solver.iterate(kk)
        self.solver.params = self.prm
        self.x_true = np.array(x0, dtype=np.float64)

    def step(self):
        """One receding-horizon step.  Returns dict(u_applied (B,2), x_true (B,6), cost (B,), K0 (B,2,6))."""
        torch = _b._torch()
        sv, B, T, dev = self.solver, self.B, self.T, self.device
        c = sv.cur
        # 1. gains about the current optimum (weights of lqr_tracking.py:324-328)
        p = self.tprob.c_problem(B, x_in_f32=1)
        st = torch.zeros(sv.Bp, dtype=torch.int32, device=dev)
        check(lib().aoc_lqr_tracking(C.byref(p), _b._ptr(sv.xb[c]), _b._ptr(sv.ub[c]), _b._ptr(sv.x0), None,
                                     _b._ptr(self.Kg), None, None, _b._ptr(st)), "aoc_lqr_tracking")
        K0 = _b.unpack_vec(self.Kg[:, 0], B).reshape(B, 2, 6)
        x_opt0 = _b.unpack_vec(sv.x0, B)                    # sample 0 of the optimum is its x0
        u_opt0 = _b.unpack_vec(sv.ub[c][:, 0], B)
        # 2. one closed-loop plant step (lqr_tracking.py:280-281) + disturbance
        xt = _b._dev_f64(self.x_true, dev)
        u_cl = u_opt0 + torch.einsum("bij,bj->bi", K0, xt - x_opt0)
        xp = _b.step_batch(self.prob.model, xt, u_cl, device=dev)[0]
        x_next = xp + self.disturbance(self.s)
        # 3. shift the input sequence and the reference window
        ub = sv.ub[c]
        ub[:, :T - 2] = ub[:, 1:T - 1].clone()              # u[T-2] repeats; u[T-1] stays 0
        self.s += 1
        self._set_window(self.s)
        # 4. warm start: x0 <- plant state, initial iterate = rollout of the shifted inputs
        sv.x0.copy_(_b.pack_vec(x_next, dev))
        n = (c + 1) % 3
        p = self.prob.c_problem(B, x_out_f32=1)
        J0 = sv.J[sv.jcur]
        check(lib().aoc_rollout_cost(C.byref(p), _b._ptr(sv.x0), _b._ptr(ub), None, None, _b._ptr(sv.xb[n]),
                                     _b._ptr(sv.ub[n]), _b._ptr(J0), _b._ptr(sv.status)), "aoc_rollout_cost")
        sv.cur, sv.cur_is64 = n, False
        # 5. re-solve: kk restarts at 0
        for kk in range(self.n_newton):
            sv.iterate(kk)
        self.x_true = x_next
        return dict(u_applied=u_cl.cpu().numpy(), x_true=x_next.copy(), cost=sv.J[sv.jcur][:B].cpu().numpy(),
                    K0=K0.cpu().numpy())
