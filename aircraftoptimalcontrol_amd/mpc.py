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

The whole step is one C-ABI call, aoc_mpc_step; host code here only advances the reference-window pointer and
draws the seeded disturbance.
"""
import ctypes as C

import numpy as np

from . import batch as _b
from ._lib import check, lib

TILE = _b.TILE


def window(ref_long, s, T):
    """Reference window [s, s+T) of a (C, >=T) curve — or (B, C, >=T) curves —, extended by holding the last sample."""
    n = ref_long.shape[-1]
    idx = np.minimum(np.arange(s, s + T), n - 1)
    return ref_long[..., idx]


class RecedingHorizon:
    def __init__(self, pr, track_weights, B, T, n_newton=2, sigma=None, seed=20260405, device="cuda:0",
                 stepsize_0=1.0, cc=0.5, beta=0.7, armijo_maxiters=10, horizon_steps=4096):
        """pr: problems.ProblemData (weights + long reference curves xx_ref (6,L), uu_ref (2,L), L >= T — or
        (B,6,L), (B,2,L): every instance tracks its own target, aoc_problem.ref_per_traj);
        track_weights: (QQt, RRt, QQT) of the tracking LQR; sigma: (6,) std of the additive state
        disturbance per closed-loop step (None = none); horizon_steps: how many steps the reference curve
        kept on the device is extended for (its last sample is held)."""
        torch = _b._torch()
        self.pr, self.B, self.T, self.n_newton = pr, int(B), int(T), int(n_newton)
        self.device = torch.device(device)
        self.tw = track_weights
        self.sigma = None if sigma is None else np.asarray(sigma, dtype=np.float64)
        self.seed = seed
        self.prm = _b.make_params(max_iters=n_newton + 1, stepsize_0=stepsize_0, cc=cc, beta=beta,
                                  armijo_maxiters=armijo_maxiters)
        self.s = 0
        # the whole (extended) reference curve lives on the device once; the window of step s is a pointer offset
        self.per_traj = np.ndim(pr.xx_ref) == 3
        n_long = max(pr.xx_ref.shape[-1], self.T) + int(horizon_steps)
        long8 = np.concatenate([window(pr.xx_ref, 0, n_long), window(pr.uu_ref, 0, n_long)], axis=-2)
        if self.per_traj:
            self.ref_long = _b.pack(long8, self.device)                                      # tiled [ntiles][n_long][8][64]
        else:
            self.ref_long = torch.from_numpy(long8.T.copy()).to(self.device)                # [n_long][8]
        self.n_long = n_long
        xr, ur = window(pr.xx_ref, 0, T), window(pr.uu_ref, 0, T)
        self.prob = _b.BatchProblem(pr.QQt, pr.RRt, pr.QQT, xr, ur, pr.dt, device=self.device)
        Q, R, QT = self.tw
        self.tprob = _b.BatchProblem(Q, R, QT, xr, ur, pr.dt, device=self.device)
        self.solver = _b.NewtonBatchSolver(self.prob, self.B, self.prm)
        self._set_window(0)
        self.x_true = None          # host copy (B,6) of the plant state
        self.x_true_d = torch.zeros_like(self.solver.x0)
        self.Kg = _b.alloc_tiled(self.B, self.T, 12, self.device)
        self.K0_d = torch.zeros((self.solver.nt, 12, TILE), dtype=torch.float64, device=self.device)
        self.ua_d = torch.zeros((self.solver.nt, 2, TILE), dtype=torch.float64, device=self.device)

    def _set_window(self, s):
        """Point both problems at the reference window [s, s+T) (a view into the device-resident curve)."""
        if s + self.T > self.n_long:
            raise ValueError("reference curve on the device exhausted: raise horizon_steps")
        for pb in (self.prob, self.tprob):
            if self.per_traj:   # the whole tiled curve array with its own sample count, the window as an element offset
                pb.ref, pb.ref_T, pb.ref_offset = self.ref_long, self.n_long, s * 8 * TILE
            else:
                pb.ref = self.ref_long[s:s + self.T]
        self.prob.xx_ref, self.prob.uu_ref = window(self.pr.xx_ref, s, self.T), window(self.pr.uu_ref, s, self.T)

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
        self.x_true_d.copy_(_b.pack_vec(self.x_true, self.device))

    def step(self, fetch=True):
        """One receding-horizon step on the device (aoc_mpc_step).  fetch: copy the per-instance results to
        the host and return dict(u_applied (B,2), x_true (B,6), cost (B,), K0 (B,2,6)); with fetch=False nothing
        is read back (the step is only enqueued) and None is returned."""
        sv, B, dev = self.solver, self.B, self.device
        c = sv.cur
        a, b = (c + 1) % 3, (c + 2) % 3
        dist = self.disturbance(self.s)
        dist_d = _b.pack_vec(dist, dev) if self.sigma is not None else None
        p_track = self.tprob.c_problem(B, x_in_f32=1, x_out_f32=1)
        self.s += 1
        self._set_window(self.s)
        p_next = self.prob.c_problem(B, x_in_f32=1, x_out_f32=1)
        slot = C.c_int32(0)
        check(lib().aoc_mpc_step(C.byref(p_track), C.byref(p_next), C.byref(self.prm), self.n_newton,
                                 _b._ptr(sv.xb[c]), _b._ptr(sv.ub[c]), _b._ptr(sv.x0), _b._ptr(self.x_true_d),
                                 _b._ptr(dist_d), _b._ptr(sv.ws), sv.ws.numel() * 8, _b._ptr(self.Kg), _b._ptr(sv.xb[a]), _b._ptr(sv.ub[a]),
                                 _b._ptr(sv.xb[b]), _b._ptr(sv.ub[b]), _b._ptr(sv.J[0]), _b._ptr(sv.J[1]),
                                 _b._ptr(sv.descent), _b._ptr(sv.stepsize), _b._ptr(sv.ntrials), _b._ptr(sv.status),
                                 _b._ptr(self.K0_d), _b._ptr(self.ua_d), C.byref(slot)), "aoc_mpc_step")
        sv.cur = a if slot.value == 0 else b
        sv.jcur, sv.cur_is64, sv.cur_rollout, sv.kk = int(slot.value), False, True, self.n_newton
        if not fetch:
            self.x_true = None
            return None
        self.x_true = _b.unpack_vec(self.x_true_d, B).cpu().numpy()
        return dict(u_applied=_b.unpack_vec(self.ua_d, B).cpu().numpy(), x_true=self.x_true.copy(),
                    cost=sv.J[sv.jcur][:B].cpu().numpy(),
                    K0=_b.unpack_vec(self.K0_d, B).reshape(B, 2, 6).cpu().numpy())
