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

The whole step is one C-ABI call, aoc_mpc_step; host code here only advances the reference-window pointer.  The
seeded disturbance is drawn ON THE DEVICE by default (aoc_mpc_noise: counter-based Philox4x32-10 keyed by (seed, global
instance index, step), Box-Muller normals): no host random numbers and no host-to-device copy on the path
of a step.  `noise_draws` below is the same generator restated in NumPy — the CHECKER of the device's draws (and what
noise="host" uploads instead, the arrangement of rounds 1-4).
"""
import ctypes as C

import numpy as np

from . import batch as _b
from ._lib import MpcNoise, check, lib

TILE = _b.TILE


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on uint32 arrays: counter (c0..c3), key (k0, k1) -> four uint32 arrays."""
    u32 = lambda a: np.asarray(a, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    c0, c1, c2, c3, k0, k1 = (u32(v) for v in (c0, c1, c2, c3, k0, k1))
    M0, M1, W0, W1, MASK = (np.uint64(v) for v in (0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF))
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & MASK, p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def noise_draws(seed, step, first, B, sigma):
    """(B,6) disturbance of closed-loop step `step` for the global instances first .. first+B-1: what the device draws
    (csrc/aoc_device.h mpc_noise_draw), restated with NumPy's log / sqrt / sin / cos (agreement ~1e-16 of sigma)."""
    inst = np.arange(first, first + B, dtype=np.uint64)
    out = np.zeros((B, 6))
    k0, k1 = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
    for j in range(3):
        w0, w1, w2, w3 = philox4x32_10(inst, np.full(B, step), np.full(B, j), np.zeros(B), np.full(B, k0), np.full(B, k1))
        uni = lambda a, b: ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64) + 1.0) * 2.0 ** -53
        u1, u2 = uni(w0, w1), uni(w2, w3)
        rad = np.sqrt(-2.0 * np.log(u1))
        out[:, 2 * j] = sigma[2 * j] * (rad * np.cos(6.283185307179586476925 * u2))
        out[:, 2 * j + 1] = sigma[2 * j + 1] * (rad * np.sin(6.283185307179586476925 * u2))
    return out


def window(ref_long, s, T):
    """Reference window [s, s+T) of a (C, >=T) curve — or (B, C, >=T) curves —, extended by holding the last sample."""
    n = ref_long.shape[-1]
    idx = np.minimum(np.arange(s, s + T), n - 1)
    return ref_long[..., idx]


class RecedingHorizon:
    def __init__(self, pr, track_weights, B, T, n_newton=2, sigma=None, seed=20260405, device="cuda:0",
                 stepsize_0=1.0, cc=0.5, beta=0.7, armijo_maxiters=10, horizon_steps=4096, noise="device", first=0):
        """pr: problems.ProblemData (weights + long reference curves xx_ref (6,L), uu_ref (2,L), L >= T — or
        (B,6,L), (B,2,L): every instance tracks its own target, aoc_problem.ref_per_traj);
        track_weights: (QQt, RRt, QQT) of the tracking LQR; sigma: (6,) std of the additive state
        disturbance per closed-loop step (None = none); horizon_steps: how many steps the reference curve
        kept on the device is extended for (its last sample is held); noise: "device" (default: the draws are made inside
        aoc_mpc_step) or "host" (the same numbers from noise_draws, uploaded per step: the checker's arrangement);
        first: global index of instance 0 (a shard of a larger fleet draws what the whole fleet would)."""
        torch = _b._torch()
        self.pr, self.B, self.T, self.n_newton = pr, int(B), int(T), int(n_newton)
        self.device = torch.device(device)
        self.tw = track_weights
        self.sigma = None if sigma is None else np.asarray(sigma, dtype=np.float64)
        self.seed, self.first = int(seed), int(first)
        if noise not in ("device", "host"):
            raise ValueError("noise must be 'device' or 'host'")
        self.noise = noise
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
        self.dist_d = torch.zeros((self.solver.nt, 6, TILE), dtype=torch.float64, device=self.device)   # what the last step added

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
        """(B,6) additive state disturbance of closed-loop step s, a function of (seed, global instance index, s): the
        host restatement of the device's generator (noise_draws)."""
        if self.sigma is None:
            return np.zeros((self.B, 6))
        return noise_draws(self.seed, s, self.first, self.B, self.sigma)

    def last_disturbance(self):
        """(B,6): what the LAST step added to the plant state, read back from the device (aoc_mpc_step's
        disturbance_out) — the exact draws, for a checker that compares plant states bit for bit."""
        return _b.unpack_vec(self.dist_d, self.B).cpu().numpy()

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
        dist_d, nz = None, None
        if self.sigma is not None:
            if self.noise == "device":
                nz = MpcNoise(self.seed, self.s, self.first, (C.c_double * 6)(*self.sigma.tolist()))
            else:
                dist_d = _b.pack_vec(self.disturbance(self.s), dev)
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
                                 _b._ptr(self.K0_d), _b._ptr(self.ua_d), C.byref(slot), C.byref(nz) if nz is not None else None,
                                 _b._ptr(self.dist_d)), "aoc_mpc_step")
        sv.cur = a if slot.value == 0 else b
        sv.jcur, sv.cur_is64, sv.cur_rollout, sv.kk = int(slot.value), False, True, self.n_newton
        if not fetch:
            self.x_true = None
            return None
        self.x_true = _b.unpack_vec(self.x_true_d, B).cpu().numpy()
        return dict(u_applied=_b.unpack_vec(self.ua_d, B).cpu().numpy(), x_true=self.x_true.copy(),
                    cost=sv.J[sv.jcur][:B].cpu().numpy(),
                    K0=_b.unpack_vec(self.K0_d, B).reshape(B, 2, 6).cpu().numpy())
