"""Batched host API over the C-ABI (include/aoc.h).

torch-ROCm is used for device memory, streams and (in bench.py) torch.distributed only; every
numerical operation of the hot path is a HIP kernel of libaoc_hip.so.  Arrays at this level use the
reference's per-trajectory conventions stacked over the batch: xx (B,6,T), uu (B,2,T), float64.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib
from ._lib import AocError, Model, Params, Problem, check, lib

TILE = _lib.AOC_TILE


def default_model(dt=1e-3):
    """Dynamics.__init__ constants (reference aircraft_simplified.py:108-118)."""
    return Model(0.1716, 2.395, 3.256, 12.0, 9.81, 0.61, 1.2, 0.24, dt)


def make_params(max_iters=200, stepsize_0=1e-2, cc=0.5, beta=0.7, armijo_maxiters=20, term_cond=-1e-6,
                hessian_switch=8):
    """NewtonMethod constructor defaults (reference optcon.py:335-339); term_cond is the value the
    reference hard-codes (-1e-6, optcon.py:368), not the ignored constructor argument."""
    return Params(int(max_iters), int(armijo_maxiters), float(stepsize_0), float(cc), float(beta),
                  float(term_cond), int(hessian_switch), 0)


def _torch():
    return _lib.require_gpu()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _dev_f64(a, device):
    torch = _torch()
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)


class BatchProblem:
    """Dynamics constants + Cost weights + reference curves on one GPU.

    Mirrors what NewtonMethod.__init__ captures (reference optcon.py:335-339).  xx_ref (6,T) / uu_ref (2,T): one
    curve shared by every trajectory of the batch (B instances of the same NewtonMethod).  xx_ref (B,6,T) / uu_ref
    (B,2,T): one curve per trajectory (B NewtonMethod instances with their own references: different manoeuvres in
    one batch, MPC instances tracking different targets); the batch size is then fixed to B."""

    def __init__(self, QQt, RRt, QQT, xx_ref, uu_ref, dt, model=None, device="cuda:0"):
        torch = _torch()
        self.device = torch.device(device)
        xx_ref = np.ascontiguousarray(xx_ref, dtype=np.float64)
        uu_ref = np.ascontiguousarray(uu_ref, dtype=np.float64)
        self.per_traj = xx_ref.ndim == 3
        if self.per_traj:
            if xx_ref.shape[1] != 6 or uu_ref.shape != (xx_ref.shape[0], 2, xx_ref.shape[2]):
                raise ValueError("per-trajectory references: xx_ref must be (B,6,T) and uu_ref (B,2,T)")
        elif xx_ref.ndim != 2 or xx_ref.shape[0] != 6 or uu_ref.shape != (2, xx_ref.shape[1]):
            raise ValueError("xx_ref must be (6,T) and uu_ref (2,T)")
        self.T = int(xx_ref.shape[-1])
        self.QQt = np.ascontiguousarray(QQt, dtype=np.float64).reshape(6, 6)
        self.RRt = np.ascontiguousarray(RRt, dtype=np.float64).reshape(2, 2)
        self.QQT = np.ascontiguousarray(QQT, dtype=np.float64).reshape(6, 6)
        self.model = model if model is not None else default_model(dt)
        self.model.dt = float(dt)
        self.xx_ref, self.uu_ref = xx_ref, uu_ref
        self.ref_T = 0          # samples per trajectory of a per-trajectory `ref` array (0 = T) and element offset of
        self.ref_offset = 0     #  the window in use: a long curve on the device, a window of it per call (mpc.py)
        if self.per_traj:
            self.B_ref = int(xx_ref.shape[0])
            self.ref = pack(np.concatenate([xx_ref, uu_ref], axis=1), self.device)    # tiled, C = 8
        else:
            self.B_ref = None
            ref = np.concatenate([xx_ref, uu_ref], axis=0).T.copy()  # [T][8]
            self.ref = torch.from_numpy(ref).to(self.device)

    def c_problem(self, B, stream=None, x_in_f32=0, x_out_f32=0, x_is_rollout=0):
        torch = _torch()
        if self.per_traj and int(B) != self.B_ref:
            raise ValueError("this problem carries %d per-trajectory reference curves; batch of %d asked" % (self.B_ref, B))
        p = Problem()
        p.x_in_f32, p.x_out_f32 = int(x_in_f32), int(x_out_f32)
        p.ref_per_traj, p.ref_T = int(self.per_traj), int(self.ref_T)
        p.x_is_rollout = int(x_is_rollout)
        p.model = self.model
        p.QQt[:] = self.QQt.ravel().tolist()
        p.RRt[:] = self.RRt.ravel().tolist()
        p.QQT[:] = self.QQT.ravel().tolist()
        p.B, p.T = int(B), self.T
        p.ref = self.ref.data_ptr() + int(self.ref_offset) * self.ref.element_size()
        s = stream if stream is not None else torch.cuda.current_stream(self.device)
        p.stream = s.cuda_stream
        return p


def ntiles(B):
    return (B + TILE - 1) // TILE


def alloc_tiled(B, T, Cc, device, zero=False, f32=False):
    torch = _torch()
    shape = (ntiles(B), T, Cc, TILE)
    return (torch.zeros if zero else torch.empty)(shape, dtype=torch.float32 if f32 else torch.float64, device=device)


def pack(a_bct, device=None, f32=False):
    """(B,C,T) fp64 -> tiled [ntiles][T][C][64] on the device (aoc_pack / aoc_pack_f32)."""
    torch = _torch()
    if device is None:
        device = a_bct.device if isinstance(a_bct, torch.Tensor) and a_bct.is_cuda else "cuda:0"
    src = _dev_f64(a_bct, device)
    B, Cc, T = src.shape
    dst = alloc_tiled(B, T, Cc, src.device, f32=f32)
    fn = lib().aoc_pack_f32 if f32 else lib().aoc_pack
    check(fn(B, T, Cc, _ptr(src), _ptr(dst), C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)), "aoc_pack")
    return dst


def unpack(tiled, B):
    """tiled (fp64 or float32) -> (B,C,T) fp64 torch tensor on the device (aoc_unpack / aoc_unpack_f32)."""
    torch = _torch()
    nt, T, Cc, _ = tiled.shape
    dst = torch.empty((B, Cc, T), dtype=torch.float64, device=tiled.device)
    fn = lib().aoc_unpack_f32 if tiled.dtype == torch.float32 else lib().aoc_unpack
    check(fn(B, T, Cc, _ptr(tiled), _ptr(dst), C.c_void_p(torch.cuda.current_stream(tiled.device).cuda_stream)),
          "aoc_unpack")
    return dst


def unpack_gains(Kt, B):
    """K~ as aoc_backward writes it — per tile and sample seven columns, the two rows of a column side by side:
    [ntiles][T][7][64][2] — -> (B,2,7,T) fp64 torch tensor on the device (the reference's KK[:, :, t], optcon.py:751)."""
    nt, T = Kt.shape[0], Kt.shape[1]
    v = Kt.reshape(nt, T, 7, TILE, 2)
    return v.permute(0, 3, 4, 2, 1).reshape(nt * TILE, 2, 7, T)[:B]


def unpack_vec(v, B):
    """[ntiles][C][64] -> (B,C)"""
    return v.permute(0, 2, 1).reshape(-1, v.shape[1])[:B]


def pack_vec(a_bc, device):
    """(B,C) per-trajectory vectors -> [ntiles][C][64]."""
    torch = _torch()
    a = _dev_f64(a_bc, device)
    B, Cc = a.shape
    nt = ntiles(B)
    if nt * TILE != B:
        a = torch.cat([a, a[-1:].expand(nt * TILE - B, Cc)], 0)
    return a.reshape(nt, TILE, Cc).permute(0, 2, 1).contiguous()


# ------------------------------------------------------------------------------------------------
# unit level
# ------------------------------------------------------------------------------------------------
def step_batch(model, x, u, lmbd=None, device="cuda:0"):
    """Dynamics.step for n points (reference aircraft_simplified.py:263-393).
    Returns numpy (xp, fx, fu, fxx, fuu, fux); the last three are None without lmbd."""
    torch = _torch()
    two_d = lambda a: a if isinstance(a, torch.Tensor) else np.atleast_2d(a)
    xd, ud = _dev_f64(two_d(x), device), _dev_f64(two_d(u), device)
    n = xd.shape[0]
    ld = None if lmbd is None else _dev_f64(two_d(lmbd), device)
    mk = lambda *s: torch.empty(s, dtype=torch.float64, device=xd.device)
    xp, fx, fu = mk(n, 6), mk(n, 6, 6), mk(n, 2, 6)
    fxx, fuu, fux = (mk(n, 6, 6), mk(n, 2, 2), mk(n, 2, 6)) if ld is not None else (None, None, None)
    st = C.c_void_p(torch.cuda.current_stream(xd.device).cuda_stream)
    check(lib().aoc_step_batch(C.byref(model), n, _ptr(xd), _ptr(ud), _ptr(ld), _ptr(xp), _ptr(fx), _ptr(fu),
                               _ptr(fxx), _ptr(fuu), _ptr(fux), st), "aoc_step_batch")
    torch.cuda.synchronize(xd.device)
    out = [xp, fx, fu, fxx, fuu, fux]
    return tuple(None if o is None else o.cpu().numpy() for o in out)


def cost_batch(problem, x, u, xr, ur):
    """Cost.stagecost/termcost for n points (reference aircraft_simplified.py:25-97).
    Returns numpy (ll, lx, lu, llT, lTx)."""
    torch = _torch()
    dev = problem.device
    xd, ud, xrd, urd = (_dev_f64(np.atleast_2d(a), dev) for a in (x, u, xr, ur))
    n = xd.shape[0]
    mk = lambda *s: torch.empty(s, dtype=torch.float64, device=dev)
    ll, lx, lu, llT, lTx = mk(n), mk(n, 6), mk(n, 2), mk(n), mk(n, 6)
    p = problem.c_problem(n)
    check(lib().aoc_cost_batch(C.byref(p), n, _ptr(xd), _ptr(ud), _ptr(xrd), _ptr(urd), _ptr(ll), _ptr(lx), _ptr(lu),
                               _ptr(llT), _ptr(lTx), C.c_void_p(p.stream)), "aoc_cost_batch")
    torch.cuda.synchronize(dev)
    return tuple(o.cpu().numpy() for o in (ll, lx, lu, llT, lTx))


# ------------------------------------------------------------------------------------------------
# iteration level
# ------------------------------------------------------------------------------------------------
class NewtonBatchSolver:
    """B independent NewtonMethod.optimize instances (reference optcon.py:341-529) on one GPU.

    Device state (tiled): three (x,u) iterate buffers in rotation — states as float32 (lossless: every
    propagated state of the reference is a float32 value, aircraft_simplified.py:300; sample 0 lives in
    the fp64 x0 array), inputs as fp64 — the K~/du workspace and per-trajectory scalars.  A
    caller-supplied initial iterate whose samples are not float32 values is kept in an extra fp64
    buffer and read from there by the first iteration.  `iterate(kk)` = one outer iteration for every
    trajectory; `solve()` adds the reference's termination and return-index behaviour per trajectory."""

    def __init__(self, problem, B, params=None, arena=None):
        """arena (vmm.Arena, optional): where the big streams of the solver live — the three iterate buffers and the
        K~/du workspace — instead of torch's allocator; `arena_layout` records what was taken.  The write-heavy streams
        (workspace first, then the iterate buffers) each start on a physical handle of their own (new_handle)."""
        torch = _torch()
        self.problem, self.B, self.T = problem, int(B), problem.T
        self.params = params if params is not None else make_params()
        dev = problem.device
        self.nt = ntiles(self.B)
        self.Bp = self.nt * TILE
        self.arena = arena
        ws_elems = (lib().aoc_workspace_bytes(self.B, self.T) + 7) // 8
        if arena is not None:
            self.ws = arena.take((ws_elems,), "float64", new_handle=True)
            self.xb = [arena.take((self.nt, self.T, 6, TILE), "float32", zero=True, new_handle=True) for _ in range(3)]
            self.ub = [arena.take((self.nt, self.T, 2, TILE), "float64", zero=True, new_handle=True) for _ in range(3)]
        else:
            self.xb = [alloc_tiled(B, self.T, 6, dev, zero=True, f32=True) for _ in range(3)]
            self.ub = [alloc_tiled(B, self.T, 2, dev, zero=True) for _ in range(3)]
            self.ws = torch.empty(ws_elems, dtype=torch.float64, device=dev)
        self.x64 = None            # fp64 copy of a caller-supplied initial iterate (only if needed)
        self.cur_is64 = False      # the current iterate is the one in self.x64
        self.cur_rollout = False   # the current iterate's states are the rollout of its inputs (written by the library)
        self.x_init = None         # (B,6,T) fp64: what set_initial() was given (returned verbatim if a
        self.u_init = None         #  trajectory stops at kk = 1, optcon.py:500-504)
        f = lambda: torch.zeros(self.Bp, dtype=torch.float64, device=dev)
        self.J = [f(), f()]
        self.descent, self.stepsize = f(), f()
        self.ntrials = torch.zeros(self.Bp, dtype=torch.int32, device=dev)
        self.status = torch.zeros(self.Bp, dtype=torch.int32, device=dev)
        self.x0 = torch.zeros((self.nt, 6, TILE), dtype=torch.float64, device=dev)
        self.spec_max = int(lib().aoc_spec_max())
        self.cur = 0      # index of the buffer holding the current iterate
        self.kk = 0       # outer-iteration index of the current iterate
        self.jcur = 0

    @property
    def n_spec(self):
        """Armijo trials riding along in the forward pass: what aoc_newton_iterate uses for this batch size under the
        aoc_tuning of the moment."""
        return int(lib().aoc_default_nspec(self.B, int(self.params.armijo_maxiters)))

    # -- problem struct with the current stream and the element types of this call's state arrays
    def _p(self, x_in_f32=1, x_out_f32=1):
        return self.problem.c_problem(self.B, x_in_f32=x_in_f32, x_out_f32=x_out_f32, x_is_rollout=self.cur_rollout)

    def _xin(self):
        """(pointer, is_f32) of the current iterate's state array"""
        return (self.x64, 0) if self.cur_is64 else (self.xb[self.cur], 1)

    def set_initial(self, xx_init, uu_init):
        """xx[:,:,0], uu[:,:,0] = xx_init, uu_init; x0 = xx_init[:,0] (reference optcon.py:395-398)."""
        torch = _torch()
        dev = self.problem.device
        xi, ui = _dev_f64(xx_init, dev), _dev_f64(uu_init, dev)
        if tuple(xi.shape) != (self.B, 6, self.T) or tuple(ui.shape) != (self.B, 2, self.T):
            raise ValueError("xx_init must be (B,6,T)=(%d,6,%d) and uu_init (B,2,T)" % (self.B, self.T))
        self.x_init, self.u_init = xi, ui
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        self.x0.copy_(pack_vec(xi[:, :, 0], dev))
        tail = xi[:, :, 1:]
        self.cur_is64 = not bool(torch.equal(tail.to(torch.float32).to(torch.float64), tail))
        if self.cur_is64:
            if self.x64 is None:
                self.x64 = alloc_tiled(self.B, self.T, 6, dev)
            check(lib().aoc_pack(self.B, self.T, 6, _ptr(xi), _ptr(self.x64), st), "aoc_pack")
        check(lib().aoc_pack_f32(self.B, self.T, 6, _ptr(xi), _ptr(self.xb[0]), st), "aoc_pack_f32")
        check(lib().aoc_pack(self.B, self.T, 2, _ptr(ui), _ptr(self.ub[0]), st), "aoc_pack")
        self.cur, self.kk, self.jcur, self.cur_rollout = 0, 0, 0, False
        self.status.zero_()
        x, f32 = self._xin()
        p = self._p(f32)
        check(lib().aoc_traj_cost(C.byref(p), _ptr(x), _ptr(self.ub[0]), _ptr(self.x0), _ptr(self.J[0])), "aoc_traj_cost")

    def set_initial_from_x0(self, x0, kp=5.0, kt=2.5):
        """Initial guess by the reference's P-controller law (aircraft_simplified.py:134-147) rolled out
        on the device from x0 (B,6), straight into the iterate buffers."""
        dev = self.problem.device
        self.x0.copy_(pack_vec(x0, dev))
        self.x_init = self.u_init = None
        self.cur_is64 = False
        p = self._p()
        check(lib().aoc_initial_trajectory(C.byref(p), float(kp), float(kt), _ptr(self.x0), _ptr(self.xb[0]),
                                           _ptr(self.ub[0])), "aoc_initial_trajectory")
        self.cur, self.kk, self.jcur, self.cur_rollout = 0, 0, 0, True
        self.status.zero_()
        check(lib().aoc_traj_cost(C.byref(p), _ptr(self.xb[0]), _ptr(self.ub[0]), _ptr(self.x0), _ptr(self.J[0])),
              "aoc_traj_cost")

    def set_initial_from_rollout(self, x0, uu):
        """Initial iterate = the open-loop rollout of uu (B,2,T) from x0 (B,6) (get_update with no direction,
        optcon.py:176-200): the warm start of BASELINE configs[2] from a saved optimum, rolled out on the device."""
        torch = _torch()
        dev = self.problem.device
        self.x0.copy_(pack_vec(x0, dev))
        ut = pack(uu, dev)
        self.x_init = self.u_init = None
        self.cur_is64 = False
        self.status.zero_()
        p = self._p()
        check(lib().aoc_rollout_cost(C.byref(p), _ptr(self.x0), _ptr(ut), None, None, _ptr(self.xb[0]), _ptr(self.ub[0]),
                                     _ptr(self.J[0]), _ptr(self.status)), "aoc_rollout_cost")
        self.cur, self.kk, self.jcur, self.cur_rollout = 0, 0, 0, True

    def iterate(self, kk=None):
        """One outer iteration (steps A-G of SURVEY 3.2) for every trajectory; asynchronous."""
        if kk is None:
            kk = self.kk
        x, f32 = self._xin()
        p = self._p(f32)
        c, n = self.cur, (self.cur + 1) % 3
        jc, jn = self.jcur, 1 - self.jcur
        check(lib().aoc_newton_iterate(C.byref(p), C.byref(self.params), int(kk), _ptr(x), _ptr(self.ub[c]),
                                       _ptr(self.x0), _ptr(self.J[jc]), _ptr(self.ws), self.ws.numel() * 8, _ptr(self.xb[n]),
                                       _ptr(self.ub[n]), _ptr(self.J[jn]), _ptr(self.descent), _ptr(self.stepsize),
                                       _ptr(self.ntrials), _ptr(self.status)), "aoc_newton_iterate")
        self.cur, self.jcur, self.kk, self.cur_is64, self.cur_rollout = n, jn, kk + 1, False, True

    PASSES = ("backward", "forward", "linesearch_search", "linesearch_update")

    def _carve(self):
        """The workspace as aoc_newton_iterate lays it out: K~ | du | trial costs | line-search scratch | candidate store
        -> (Kt, du, J_trial, scratch, scratch bytes, n_spec, cand or None, cand bytes).  n_spec and the number of stored
        candidates are asked from the library at call time (aoc_default_nspec / aoc_default_ncand follow aoc_tuning), and
        the candidate store is handed over only if the workspace has room for it — as aoc_newton_iterate decides."""
        l = lib()
        nel14 = l.aoc_tiled_elems(self.B, self.T, 14)
        nel2 = l.aoc_tiled_elems(self.B, self.T, 2)
        Kt, du = self.ws[:nel14], self.ws[nel14:nel14 + nel2]
        o = nel14 + nel2
        Jt = self.ws[o:o + self.spec_max * self.Bp]
        o += self.spec_max * self.Bp
        scratch = self.ws[o:]
        sbytes = int(l.aoc_linesearch_scratch_bytes(self.B, self.T))
        sb = (sbytes + 255) // 256 * 256
        nsp = int(l.aoc_default_nspec(self.B, int(self.params.armijo_maxiters)))
        ncand = int(l.aoc_default_ncand(self.B, nsp, int(self.params.armijo_maxiters)))
        cbytes = int(l.aoc_candidate_bytes(self.B, self.T, ncand)) if ncand else 0
        cand = None
        if ncand and sb + cbytes <= scratch.numel() * 8:
            cand = scratch[sb // 8:]
        else:
            cbytes = 0
        return Kt, du, Jt, scratch, sbytes, nsp, cand, cbytes

    def iterate_passes(self, kk=None, between=None):
        """The launches of iterate(), issued pass by pass through the four entry points aoc_newton_iterate is made of:
        backward | forward | line-search rounds (aoc_linesearch_search) | final rollout (aoc_linesearch_update).
        between(i), if given, is called on the host before pass i (i indexing PASSES) and once more, with i = 4, after
        the last one — where a caller records events on the launch stream or makes it wait for another stream."""
        torch = _torch()
        if kk is None:
            kk = self.kk
        x, f32 = self._xin()
        p = self._p(f32)
        prm = self.params
        c, n = self.cur, (self.cur + 1) % 3
        jc, jn = self.jcur, 1 - self.jcur
        Kt, du, Jt, scratch, sbytes, nsp, cand, cbytes = self._carve()
        between = between or (lambda i: None)
        between(0)
        # (what lies behind K~ in the workspace is free until the forward pass: scratch of the backward pass, as in aoc_newton_iterate)
        check(lib().aoc_backward(C.byref(p), int(kk > prm.hessian_switch), _ptr(x), _ptr(self.ub[c]), _ptr(self.x0),
                                 _ptr(Kt), None, _ptr(self.status), _ptr(du), (self.ws.numel() - Kt.numel()) * 8), "aoc_backward")
        between(1)
        check(lib().aoc_forward(C.byref(p), C.byref(prm), nsp, _ptr(x), _ptr(self.ub[c]), _ptr(self.x0),
                                _ptr(Kt), _ptr(du), _ptr(self.descent), _ptr(Jt), _ptr(self.status), _ptr(cand), cbytes,
                                _ptr(self.ntrials)), "aoc_forward")
        between(2)
        check(lib().aoc_linesearch_search(C.byref(p), C.byref(prm), nsp, _ptr(self.ub[c]), _ptr(self.x0), _ptr(du),
                                          _ptr(self.J[jc]), _ptr(self.descent), _ptr(Jt), _ptr(self.stepsize),
                                          _ptr(self.ntrials), _ptr(scratch), sbytes), "aoc_linesearch_search")
        between(3)
        check(lib().aoc_linesearch_update(C.byref(p), C.byref(prm), _ptr(self.ub[c]), _ptr(self.x0), _ptr(du),
                                          _ptr(self.xb[n]), _ptr(self.ub[n]), _ptr(self.J[jn]), _ptr(self.stepsize),
                                          _ptr(self.ntrials), _ptr(self.status), _ptr(scratch), sbytes, nsp, _ptr(Jt),
                                          _ptr(cand), cbytes), "aoc_linesearch_update")
        between(4)
        self.cur, self.jcur, self.kk, self.cur_is64, self.cur_rollout = n, jn, kk + 1, False, True

    def iterate_timed(self, kk=None):
        """iterate_passes() with HIP events recorded on the launch stream between the passes.  Returns the five events;
        read them after a synchronize with ev[i].elapsed_time(ev[i+1]) [ms], i indexing PASSES."""
        torch = _torch()
        st = torch.cuda.current_stream(self.problem.device)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        self.iterate_passes(kk, lambda i: ev[i].record(st))
        return ev

    # -- results ---------------------------------------------------------------------------------
    def direction(self):
        """du (B,2,T) of the last iteration (it stays in the workspace until the next one)."""
        nel14 = lib().aoc_tiled_elems(self.B, self.T, 14)
        nel2 = lib().aoc_tiled_elems(self.B, self.T, 2)
        return unpack(self.ws[nel14:nel14 + nel2].view(self.nt, self.T, 2, TILE), self.B).cpu().numpy()

    def previous_inputs(self):
        """uu (B,2,T) of the iterate the last iteration started from (the buffers rotate over three slots)."""
        return unpack(self.ub[(self.cur - 1) % 3], self.B).cpu().numpy()

    def _unpack_iterate(self, i):
        """(B,6,T), (B,2,T) fp64 device tensors of buffer i, sample 0 restored from the fp64 x0."""
        if self.cur_is64 and i == self.cur:
            xx = unpack(self.x64, self.B)
        else:
            xx = unpack(self.xb[i], self.B)
            xx[:, :, 0] = unpack_vec(self.x0, self.B)
        return xx, unpack(self.ub[i], self.B)

    def current(self):
        """Current iterate as numpy (B,6,T), (B,2,T)."""
        xx, uu = self._unpack_iterate(self.cur)
        return xx.cpu().numpy(), uu.cpu().numpy()

    def scalars(self):
        """Per-trajectory scalars of the LAST iteration: cost of the iterate it started from, descent,
        accepted step, number of Armijo trials, cost of the new iterate, status flags."""
        B = self.B
        return dict(cost=self.J[1 - self.jcur][:B].cpu().numpy(), descent=self.descent[:B].cpu().numpy(),
                    stepsize=self.stepsize[:B].cpu().numpy(), ntrials=self.ntrials[:B].cpu().numpy(),
                    cost_new=self.J[self.jcur][:B].cpu().numpy(), status=self.status[:B].cpu().numpy())

    def join(self):
        """(interface shared with TwoStreamNewtonSolver: one stream, nothing to wait for)"""

    def summary_tensors(self):
        """(cost of the newest iterate, descent, trial count) of every trajectory as device tensors."""
        B = self.B
        return self.J[self.jcur][:B], self.descent[:B], self.ntrials[:B]

    def summary(self, out=None, accumulate=False):
        """sharding.SUMMARY_FIELDS of this batch as five fp64 scalars on the device, by aoc_summary (one kernel, fixed
        reduction order); accumulate: added to what `out` holds.  What the path's one collective reduces."""
        torch = _torch()
        dev = self.problem.device
        if out is None:
            out = torch.zeros(5, dtype=torch.float64, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        check(lib().aoc_summary(self.B, _ptr(self.J[self.jcur]), _ptr(self.descent), _ptr(self.ntrials), _ptr(out),
                                int(bool(accumulate)), st), "aoc_summary")
        return out

    def run_fixed(self, n_iters, kk0=None, record=True):
        """n_iters iterations for every trajectory, no early exit (the bench mode)."""
        hist = []
        if kk0 is not None:
            self.kk = kk0
        for _ in range(n_iters):
            self.iterate()
            if record:
                hist.append(self.scalars())
        return hist

    def _compacted(self, keep):
        """A new solver holding only the trajectories `keep` (device index tensor into this one): their
        current and previous iterate, x0, current cost and status."""
        torch = _torch()
        n, dev = self.B, self.problem.device
        m = int(keep.numel())
        nw = type(self)(self.problem, m, self.params)   # a GradientBatchSolver stays one (its iterate() differs)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for src, dst in (((self.cur - 1) % 3, 0), (self.cur, 1)):   # previous -> slot 0, current -> slot 1
            xs = unpack(self.xb[src], n)[keep].contiguous()
            us = unpack(self.ub[src], n)[keep].contiguous()
            check(lib().aoc_pack_f32(m, self.T, 6, _ptr(xs), _ptr(nw.xb[dst]), st), "aoc_pack_f32")
            check(lib().aoc_pack(m, self.T, 2, _ptr(us), _ptr(nw.ub[dst]), st), "aoc_pack")
        nw.x0.copy_(pack_vec(unpack_vec(self.x0, n)[keep], dev))
        nw.J[0][:m] = self.J[self.jcur][:n][keep]
        nw.status[:m] = self.status[:n][keep]
        if self.x_init is not None:
            nw.x_init, nw.u_init = self.x_init[keep], self.u_init[keep]
        nw.cur, nw.jcur, nw.kk, nw.cur_is64, nw.cur_rollout = 1, 0, self.kk, False, True
        return nw

    def solve(self, verbose=False, callback=None, compact=True, compact_below=0.5, compact_min=2048):
        """NewtonMethod.optimize semantics per trajectory (reference optcon.py:415-505):
        iterate kk = 0..max_iters-2; a trajectory stops at the first kk with descent >= term_cond and
        then returns iterate kk-1 (python index -1 = an all-zero history slot when kk == 0); without
        convergence it returns the last computed iterate; finally uu_star[:,-1] = uu_star[:,-2].
        compact: once fewer than `compact_below` of the trajectories in flight are still iterating (and
        more than `compact_min` are in flight) the active ones are re-packed into a smaller batch, so
        that converged trajectories stop costing passes; results do not depend on it.
        Returns dict(xx_star (B,6,T), uu_star (B,2,T), iters (B,), converged, status, history of per-iteration
        scalars (B,n_iter), NaN / -1 where a trajectory was no longer iterating)."""
        torch = _torch()
        B, prm, T = self.B, self.params, self.T
        dev = self.problem.device
        out_x = torch.zeros((B, 6, T), dtype=torch.float64, device=dev)
        out_u = torch.zeros((B, 2, T), dtype=torch.float64, device=dev)
        g_iters = torch.zeros(B, dtype=torch.int32, device=dev)
        g_conv = torch.zeros(B, dtype=torch.bool, device=dev)
        g_status = torch.zeros(B, dtype=torch.int32, device=dev)
        hist = dict(cost=[], descent=[], stepsize=[], ntrials=[])

        def fresh(sv):
            act = torch.zeros(sv.Bp, dtype=torch.bool, device=dev)
            act[:sv.B] = True
            return dict(active=act, ret=torch.full((sv.Bp,), -2, dtype=torch.int32, device=dev),
                        res_x=torch.zeros_like(sv.xb[0]), res_u=torch.zeros_like(sv.ub[0]))

        def flush(sv, gidx, L, final_kk=None):
            """write the results of this generation's finished trajectories into the global arrays"""
            n = sv.B
            if final_kk is not None:  # still iterating at the end: the last computed iterate (index max_iters-1)
                m = L["active"].reshape(sv.nt, 1, 1, TILE)
                L["res_x"] = torch.where(m, sv.xb[sv.cur], L["res_x"])
                L["res_u"] = torch.where(m, sv.ub[sv.cur], L["res_u"])
                L["ret"] = torch.where(L["active"], torch.full_like(L["ret"], final_kk + 1), L["ret"])
            ret = L["ret"][:n]
            done = ret != -2
            if bool(done.any()):
                xs, us = unpack(L["res_x"], n), unpack(L["res_u"], n)
                real = ret >= 0
                xs[:, :, 0] = torch.where(real[:, None], unpack_vec(sv.x0, n), xs[:, :, 0])  # sample 0 is the fp64 x0
                if sv.x_init is not None:  # index 0 is the caller's initial iterate, verbatim
                    is0 = ret == 0
                    xs = torch.where(is0[:, None, None], sv.x_init, xs)
                    us = torch.where(is0[:, None, None], sv.u_init, us)
                us[:, :, -1] = us[:, :, -2]  # optcon.py:505
                tgt = gidx[done]
                out_x[tgt], out_u[tgt] = xs[done], us[done]
            if final_kk is not None:  # status of the trajectories that never stopped
                live = L["active"][:n]
                g_status[gidx[live]] = sv.status[:n][live]

        sv, gidx = self, torch.arange(B, device=dev)
        L = fresh(sv)
        last = -1
        for kk in range(prm.max_iters - 1):
            sv.iterate(kk)
            last = kk
            n = sv.B
            sc = sv.scalars()
            act = L["active"]
            was = act[:n].cpu().numpy()                 # iterating at the start of this iteration
            gnp = gidx.cpu().numpy()[was]
            for k_ in hist:
                full = np.full(B, -1, dtype=np.int64) if k_ == "ntrials" else np.full(B, np.nan)
                full[gnp] = sc[k_][was]
                hist[k_].append(full)
            conv = act & (sv.descent >= prm.term_cond)
            g_iters[gidx[act[:n]]] = kk + 1
            if bool(conv.any()):
                m = conv.reshape(sv.nt, 1, 1, TILE)
                if kk >= 1:  # iterate kk-1 lives two buffers behind the newest
                    L["res_x"] = torch.where(m, sv.xb[(sv.cur + 1) % 3], L["res_x"])
                    L["res_u"] = torch.where(m, sv.ub[(sv.cur + 1) % 3], L["res_u"])
                # kk == 0: xx[:,:,-1] is the untouched, all-zero last history slot
                L["ret"] = torch.where(conv, torch.full_like(L["ret"], kk - 1), L["ret"])
                sv.status |= conv.to(torch.int32) * _lib.ST_CONVERGED
                g_conv[gidx[conv[:n]]] = True
                g_status[gidx[conv[:n]]] = sv.status[:n][conv[:n]]  # events up to the stopping iteration
                L["active"] = act = act & ~conv
            if callback is not None:
                callback(kk, sc)
            n_act = int(act.sum())
            if verbose:
                print("Iter = %d\t in flight = %d\t still iterating = %d" % (kk, n, n_act))
            if n_act == 0:
                break
            if compact and n > compact_min and n_act < compact_below * n and kk < prm.max_iters - 2:
                keep = torch.nonzero(act[:n]).squeeze(1)
                flush(sv, gidx, L)
                sv, gidx = sv._compacted(keep), gidx[keep]
                L = fresh(sv)
        flush(sv, gidx, L, final_kk=last)
        return dict(xx_star=out_x.cpu().numpy(), uu_star=out_u.cpu().numpy(), iters=g_iters.cpu().numpy(),
                    converged=g_conv.cpu().numpy(), status=g_status.cpu().numpy(),
                    history={k_: np.stack(v, 1) if v else np.zeros((B, 0)) for k_, v in hist.items()},
                    last_kk=last)

    def solve_on_device(self, sync_every=4, history=True, x_star_f64=False, to_host=True, two_streams=True):
        """solve() through aoc_newton_solve2: the whole loop, stopping rule and return-index bookkeeping on
        the device; the host only reads the count of still-iterating trajectories (every `sync_every` iterations
        while nobody has stopped, more often afterwards; 0: never, all max_iters-1 iterations run) to stop launching
        and to re-pack the iterating trajectories into denser generations.  two_streams: large batches (aoc_tuning.
        solve_split_tiles) are cut in two halves on two HIP streams that never wait for each other.
        Same return value as solve(); results are identical to solve(compact=False).
        to_host=False leaves xx_star (B,6,T) / uu_star (B,2,T) on the device as torch tensors (copying 65 536
        trajectories to the host takes longer than solving them)."""
        torch = _torch()
        B, T, prm, dev = self.B, self.T, self.params, self.problem.device
        n_it = max(int(prm.max_iters) - 1, 0)
        x, f32 = self._xin()
        p = self._p(f32, 0 if x_star_f64 else 1)   # x_star as fp64 or float32 (the same values for iterates >= 1)
        ws = torch.empty((lib().aoc_solve_workspace_bytes(B, T) + 7) // 8, dtype=torch.float64, device=dev)
        x_star = alloc_tiled(B, T, 6, dev, f32=not x_star_f64)
        u_star = torch.zeros_like(self.ub[0])
        iters = torch.zeros(self.Bp, dtype=torch.int32, device=dev)
        ret = torch.zeros(self.Bp, dtype=torch.int32, device=dev)
        status = torch.zeros(self.Bp, dtype=torch.int32, device=dev)
        hf = lambda dt: torch.empty((max(n_it, 1), self.Bp), dtype=dt, device=dev) if history else None
        hc, hd, hs, hn = hf(torch.float64), hf(torch.float64), hf(torch.float64), hf(torch.int32)
        n_run = C.c_int32(0)
        st2 = None
        if two_streams:   # a stream that runs beside the CURRENT one (the pairing is checked once per current stream)
            cur = torch.cuda.current_stream(dev)
            pairs = self.__dict__.setdefault("_stream2_for", {})
            if cur.cuda_stream not in pairs:
                pairs[cur.cuda_stream] = concurrent_streams(dev, 2, first=cur)[1]
            st2 = C.c_void_p(pairs[cur.cuda_stream].cuda_stream)
        torch.cuda.synchronize(dev)
        t_start = time.perf_counter()
        check(lib().aoc_newton_solve2(C.byref(p), C.byref(prm), _ptr(x), _ptr(self.ub[self.cur]), _ptr(self.x0), _ptr(ws), ws.numel() * 8,
                                      int(sync_every), _ptr(x_star), _ptr(u_star), _ptr(iters), _ptr(ret), _ptr(status),
                                      _ptr(hc), _ptr(hd), _ptr(hs), _ptr(hn), C.byref(n_run), st2), "aoc_newton_solve2")
        torch.cuda.synchronize(dev)
        t_device = time.perf_counter() - t_start   # the solve itself; the rest is unpacking and the copy to the host
        del ws
        r = ret[:B]
        xs, us = unpack(x_star, B), unpack(u_star, B)
        xs[:, :, 0] = torch.where((r >= 0)[:, None], unpack_vec(self.x0, B), xs[:, :, 0])  # sample 0 is the fp64 x0
        if self.x_init is not None:  # index 0 is the caller's initial iterate, verbatim
            is0 = r == 0
            xs = torch.where(is0[:, None, None], self.x_init, xs)
            us = torch.where(is0[:, None, None], self.u_init, us)
            us[:, :, -1] = torch.where(is0[:, None], us[:, :, -2], us[:, :, -1])  # optcon.py:505
        st = status[:B].cpu().numpy()
        n = int(n_run.value)
        hist = {}
        for key, h in (("cost", hc), ("descent", hd), ("stepsize", hs), ("ntrials", hn)):
            hist[key] = h[:n, :B].T.cpu().numpy() if history else np.zeros((B, 0))
        return dict(xx_star=xs.cpu().numpy() if to_host else xs, uu_star=us.cpu().numpy() if to_host else us,
                    iters=iters[:B].cpu().numpy(), converged=(st & _lib.ST_CONVERGED) != 0, status=st, history=hist,
                    last_kk=n - 1, device_seconds=t_device)


def solver_arena_bytes(B, T, gran=1 << 21, chunk=None):
    """Bytes an arena must have for one NewtonBatchSolver(B, T) (seven buffers, each rounded up to a handle boundary)."""
    a = chunk or gran
    up = lambda v: (int(v) + a - 1) // a * a
    nt = ntiles(B)
    return (up(lib().aoc_workspace_bytes(B, T) + 8) + 3 * up(nt * T * 6 * TILE * 4) + 3 * up(nt * T * 2 * TILE * 8) + a)


class GradientBatchSolver(NewtonBatchSolver):
    """B independent GradientMethod.optimize instances (reference optcon.py:27-174): steepest descent,
    du = -(B^T lambda + l_u) from the costate sweep (aoc_gradient), Armijo back-tracking and update through the same
    line search as the Newton iteration (aoc_linesearch with no speculated trials).  The reference's own method cannot
    run (TypeError at optcon.py:125); this is its loop with the missing JP argument supplied — parity unpinned,
    checked against the oracle's restatement only.  `descent` holds the slope -sum |du|^2 (the reference prints
    +sum |du|^2 and stops when that is <= 1e-6: the same test as slope >= term_cond = -1e-6, so solve() applies
    unchanged, return-index behaviour included)."""

    def iterate(self, kk=None):
        if kk is None:
            kk = self.kk
        x, f32 = self._xin()
        p = self._p(f32)
        c, n = self.cur, (self.cur + 1) % 3
        jc, jn = self.jcur, 1 - self.jcur
        _, du, _, scratch, sbytes, _, _, _ = self._carve()
        check(lib().aoc_gradient(C.byref(p), _ptr(x), _ptr(self.ub[c]), _ptr(self.x0), _ptr(du), _ptr(self.descent),
                                 _ptr(self.status)), "aoc_gradient")
        check(lib().aoc_linesearch(C.byref(p), C.byref(self.params), 0, _ptr(self.ub[c]), _ptr(self.x0), _ptr(du),
                                   _ptr(self.J[jc]), _ptr(self.descent), None, _ptr(self.xb[n]), _ptr(self.ub[n]),
                                   _ptr(self.J[jn]), _ptr(self.stepsize), _ptr(self.ntrials), _ptr(self.status),
                                   _ptr(scratch), sbytes, None, 0), "aoc_linesearch")
        self.cur, self.jcur, self.kk, self.cur_is64, self.cur_rollout = n, jn, kk + 1, False, True

    def direction(self):
        """du (B,2,T) of the last iteration."""
        nel14 = lib().aoc_tiled_elems(self.B, self.T, 14)
        nel2 = lib().aoc_tiled_elems(self.B, self.T, 2)
        return unpack(self.ws[nel14:nel14 + nel2].view(self.nt, self.T, 2, TILE), self.B).cpu().numpy()

    def solve_on_device(self, *a, **kw):
        raise NotImplementedError("aoc_newton_solve runs the Newton iteration; use solve()")


def concurrent_streams(device, n=2, first=None, attempts=16):
    """n HIP streams that really run side by side.  The runtime maps streams onto a few hardware queues (four by default)
    and streams that share a queue take turns: of eight streams created one after the other {2, 3, 7}, {0, 5} and {1, 4}
    shared one, and a two-stream solver on such a pair runs at one-stream speed (5.7 instead of 4.6-4.95 ms per
    iteration: the "second candidate is always the slowest" of the placement probes of round 3 was this).  New streams
    are created until n of them are pairwise concurrent (aoc_streams_concurrent: a 0.2 ms kernel on each).  `first`: a
    stream that must be among them (e.g. the caller's current stream).  Falls back to what it has after `attempts`."""
    torch = _torch()
    dev = torch.device(device)
    good = [first] if first is not None else []
    spare = []
    with torch.cuda.device(dev):      # the probe's events are created on the current device: the streams' device
        for _ in range(attempts):
            if len(good) >= n:
                break
            st = torch.cuda.Stream(device=dev)
            verdicts = [lib().aoc_streams_concurrent(C.c_void_p(g.cuda_stream), C.c_void_p(st.cuda_stream)) for g in good]
            for v in verdicts:
                if v < 0:
                    check(v, "aoc_streams_concurrent")
            (good if all(v == 1 for v in verdicts) else spare).append(st)
        while len(good) < n:      # not enough distinct queues: take what there is
            good.append(spare.pop() if spare else torch.cuda.Stream(device=dev))
    return good


class TwoStreamNewtonSolver:
    """Fixed-iteration runs of a large batch as two half batches on two HIP streams, each a NewtonBatchSolver of its
    own: every kernel takes its stream from aoc_problem, so nothing in the library changes — the halves simply never
    wait for each other.  While one half sits in its latency-bound line-search rounds the other streams through HBM
    (measured, 131 072 trajectories, T = 500: 5.87 -> 5.36 ms per iteration; cutting the batch in two INSIDE one call,
    with a join at the end of every iteration, was slower than one stream: half-size launches only pay while the other
    half keeps the rest of the chip busy).  Per-trajectory results do not depend on the batch a trajectory is solved
    in, so they equal those of one NewtonBatchSolver bit for bit.
    Interface: set_initial_from_x0 / iterate / run_fixed / join / scalars / current / summary_tensors."""

    def __init__(self, problem, B, params=None, streams=None, arenas=None):
        torch = _torch()
        nt = ntiles(B)
        if nt < 2:
            raise ValueError("a batch of one tile cannot be cut in two")
        self.problem, self.B, self.T = problem, int(B), problem.T
        self.Ba = (nt // 2) * TILE
        ar = arenas if arenas is not None else (None, None)      # vmm.Arena per half (see NewtonBatchSolver)
        self.parts = [NewtonBatchSolver(problem, self.Ba, params, arena=ar[0]),
                      NewtonBatchSolver(problem, self.B - self.Ba, params, arena=ar[1])]
        self.params = self.parts[0].params
        # two streams on ONE hardware queue would take turns (concurrent_streams): checked once, here
        self.streams = list(streams) if streams is not None else concurrent_streams(problem.device, len(self.parts))
        self.kk = 0
        self.phased = bool(int(os.environ.get("AOC_TWO_STREAM_PHASED", "0")))   # iterate_phased as the schedule of iterate()
        self._search_done = [None, None]

    def _on(self, fn):
        torch = _torch()
        cur = torch.cuda.current_stream(self.problem.device)
        for i, (sv, st) in enumerate(zip(self.parts, self.streams)):
            st.wait_stream(cur)                      # inputs prepared on the caller's stream are ready
            with torch.cuda.stream(st):
                fn(i, sv)

    def join(self):
        """The caller's stream waits for both halves (before anything reads their results)."""
        torch = _torch()
        cur = torch.cuda.current_stream(self.problem.device)
        for st in self.streams:
            cur.wait_stream(st)

    def set_initial_from_x0(self, x0, kp=5.0, kt=2.5):
        x0 = _dev_f64(x0, self.problem.device)
        cut = (x0[:self.Ba], x0[self.Ba:])
        self._on(lambda i, sv: sv.set_initial_from_x0(cut[i], kp, kt))
        self.kk = 0
        self._search_done = [None, None]
        self.join()

    def iterate(self, kk=None):
        """One outer iteration of every trajectory: enqueued on the two streams, no join (`phased`, off by default:
        see iterate_phased)."""
        if kk is None:
            kk = self.kk
        if self.phased:
            return self.iterate_phased(kk)
        torch = _torch()
        for sv, st in zip(self.parts, self.streams):
            with torch.cuda.stream(st):
                sv.iterate(kk)
        self.kk = kk + 1

    def iterate_phased(self, kk=None):
        """iterate() with the two halves taking turns in their line-search rounds: a half's search waits
        (hipStreamWaitEvent) for the end of the other half's previous search, so the order on the device is A0 B0 A1
        B1 ... and every search runs beside the other half's update / backward / forward passes, never beside its
        search (left to themselves only 40-60 % of a half's search time lies beside a streaming pass of the other one,
        profiles/r05_head_overlap_two_streams.json).  Scheduling only: the same four entry points per half and iteration
        as aoc_newton_iterate issues, results identical (tests/test_gpu_multirank.py).  MEASURED SLOWER and therefore
        off (AOC_TWO_STREAM_PHASED=1 or .phased turns it on): 5.09 against 5.01 ms per iteration over kk 0..19, 4.73 / 4.58
        over kk 0..9 (tools/phase_probe.py, one set of buffers) — the search is not idle time of the chip that a
        schedule could fill: a half's search takes 0.86-1.26 ms per iteration beside the other half (the whole batch's on one
        stream: 1.05 ms) and the passes beside it stretch (profiles/r05_head_overlap_two_streams.json, r05_sq_counters.txt)
        (EXPERIMENTS.md "Two streams, the search as a token")."""
        if kk is None:
            kk = self.kk
        torch = _torch()
        for i, (sv, st) in enumerate(zip(self.parts, self.streams)):
            def between(j, i=i, st=st):
                if j == 2 and self._search_done[1 - i] is not None:
                    st.wait_event(self._search_done[1 - i])
                elif j == 3:
                    ev = torch.cuda.Event()
                    ev.record(st)
                    self._search_done[i] = ev
            with torch.cuda.stream(st):
                sv.iterate_passes(kk, between)
        self.kk = kk + 1

    def run_fixed(self, n_iters, kk0=None):
        if kk0 is not None:
            self.kk = kk0
        for _ in range(n_iters):
            self.iterate()
        self.join()

    def scalars(self):
        self.join()
        a, b = (sv.scalars() for sv in self.parts)
        return {k: np.concatenate([a[k], b[k]]) for k in a}

    def current(self):
        self.join()
        (xa, ua), (xb, ub) = (sv.current() for sv in self.parts)
        return np.concatenate([xa, xb]), np.concatenate([ua, ub])

    def summary_tensors(self):
        """(cost of the newest iterate, descent, trial count) of every trajectory as device tensors, after a join."""
        torch = _torch()
        self.join()
        cat = lambda f: torch.cat([f(sv)[:sv.B] for sv in self.parts])
        return cat(lambda sv: sv.J[sv.jcur]), cat(lambda sv: sv.descent), cat(lambda sv: sv.ntrials)

    def summary(self, out=None):
        """sharding.SUMMARY_FIELDS of the whole batch on the device (aoc_summary: first half, then the second half
        accumulated), on the caller's stream after a join."""
        self.join()
        out = self.parts[0].summary(out)
        return self.parts[1].summary(out, accumulate=True)


def best_placed(make_solver, x0, candidates=5, probe_iters=6, keep_first=False, force=None):
    """Allocation autotuning for large fixed-iteration runs: build `candidates` solvers (make_solver() -> a
    NewtonBatchSolver or TwoStreamNewtonSolver, each with allocations of its own), time Newton iterations
    kk = 0..probe_iters-1 from x0 (B,6) on each, keep the fastest and free the others.

    Why: on MI355X the duration of the write-heavy passes depends on which physical pages hold their output buffers —
    per launch of the backward pass the L2's write requests stall 4-5 times longer for DRAM credits on a "slow"
    allocation, for the same bytes (profiles/r03_placement_pmc.txt) — and the effect is a property of the allocation,
    stable for its lifetime: five two-stream solvers built one after the other in one process run the same 20
    iterations in 4.96, 5.07, 5.47, 5.51 and 6.15 ms each, every time (tools/two_stream_lottery.py).  All candidates
    stay alive until the choice is made (a freed block would simply be handed out again).  Results do not depend on
    the choice: the buffers hold the same values wherever they are.
    Returns (solver, info) with info = {"ms_per_iteration": [...], "chosen": index, "probe_iterations", "probe_wall_s":
    what the choice cost}.  keep_first: candidate 0 — the allocation a caller who does not choose gets — stays alive
    too and is returned as info["first"] (the chosen solver itself when that is candidate 0), so that it can be timed on
    the same workload.  force (tests): take candidate `force % candidates` whatever the probe says."""
    torch = _torch()
    t_wall = time.perf_counter()
    if int(candidates) <= 1:          # nothing to choose from: no probe either
        sv = make_solver()
        info = {"ms_per_iteration": [], "chosen": 0, "probe_iterations": 0, "probe_wall_s": 0.0}
        if keep_first:
            info["first"] = sv
        return sv, info
    cands, ms = [], []
    for _ in range(int(candidates)):
        sv = make_solver()
        dev = sv.problem.device
        xd = _dev_f64(x0, dev)
        best = None
        for rep in range(2):                      # the first pass also warms up lazily loaded code objects
            sv.set_initial_from_x0(xd)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for kk in range(probe_iters):
                sv.iterate(kk)
            sv.join()
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t0) / probe_iters * 1e3
            best = dt if best is None else min(best, dt)
        cands.append(sv)
        ms.append(round(best, 3))
    chosen = int(np.argmin(ms)) if force is None else int(force) % len(cands)
    keep, first = cands[chosen], cands[0]
    del cands, sv
    info = {"ms_per_iteration": ms, "chosen": chosen, "probe_iterations": probe_iters}
    if keep_first:
        info["first"] = first
    del first
    torch.cuda.empty_cache()
    info["probe_wall_s"] = round(time.perf_counter() - t_wall, 3)
    return keep, info


def traj_cost(problem, xx, uu):
    """Cost of stored trajectories (reference optcon.py:417-424).  xx (B,6,T), uu (B,2,T) -> (B,)"""
    torch = _torch()
    xx = np.asarray(xx, dtype=np.float64)
    xt, ut = pack(xx, problem.device), pack(uu, problem.device)
    B = xx.shape[0]
    x0t = pack_vec(xx[:, :, 0], problem.device)
    J = torch.empty(ntiles(B) * TILE, dtype=torch.float64, device=problem.device)
    p = problem.c_problem(B)
    check(lib().aoc_traj_cost(C.byref(p), _ptr(xt), _ptr(ut), _ptr(x0t), _ptr(J)), "aoc_traj_cost")
    return J[:B].cpu().numpy()


def rollout_cost(problem, x0, uu, du=None, alpha=None, write=True, f32=False):
    """get_update + trial cost (reference optcon.py:176-200, :250-264).
    x0 (B,6), uu (B,2,T), du (B,2,T) or None, alpha (B,) -> (xx (B,6,T), uu' (B,2,T), J (B,), status).
    f32: store the state trajectory as float32 on the device (same values)."""
    torch = _torch()
    dev = problem.device
    B = uu.shape[0]
    ut = pack(uu, dev)
    dt_ = None if du is None else pack(du, dev)
    x0t = pack_vec(x0, dev)
    nt = ntiles(B)
    al = None
    if du is not None:
        al = torch.zeros(nt * TILE, dtype=torch.float64, device=dev)
        al[:B] = _dev_f64(np.broadcast_to(np.asarray(alpha, dtype=np.float64), (B,)).copy(), dev)
    xo = alloc_tiled(B, problem.T, 6, dev, f32=f32) if write else None
    uo = alloc_tiled(B, problem.T, 2, dev) if write else None
    J = torch.empty(nt * TILE, dtype=torch.float64, device=dev)
    st = torch.zeros(nt * TILE, dtype=torch.int32, device=dev)
    p = problem.c_problem(B, x_out_f32=int(f32))
    check(lib().aoc_rollout_cost(C.byref(p), _ptr(x0t), _ptr(ut), _ptr(dt_), _ptr(al), _ptr(xo), _ptr(uo), _ptr(J),
                                 _ptr(st)), "aoc_rollout_cost")
    xx = un = None
    if write:
        xd = unpack(xo, B)
        xd[:, :, 0] = _dev_f64(x0, dev)
        xx, un = xd.cpu().numpy(), unpack(uo, B).cpu().numpy()
    return xx, un, J[:B].cpu().numpy(), st[:B].cpu().numpy()


def backward_forward(problem, xx, uu, full_hessian, stepsize_0=1.0, f32=False):
    """One backward + forward pass (reference optcon.py:429-477 and the first Armijo trial).
    Returns dict with K~ (B,2,7,T), du (B,2,T), descent (B,), lmbd0 (B,6), J_trial0, status.
    f32: hand the state trajectory over as float32 (requires float32-valued samples t >= 1)."""
    torch = _torch()
    dev = problem.device
    xx = np.asarray(xx, dtype=np.float64)
    B, T = xx.shape[0], problem.T
    nt = ntiles(B)
    xt, ut = pack(xx, dev, f32=f32), pack(uu, dev)
    Kt = alloc_tiled(B, T, 14, dev, zero=True)
    du = alloc_tiled(B, T, 2, dev)
    lm0 = torch.empty((nt, 6, TILE), dtype=torch.float64, device=dev)
    desc = torch.empty(nt * TILE, dtype=torch.float64, device=dev)
    Jn = torch.empty(nt * TILE, dtype=torch.float64, device=dev)
    st = torch.zeros(nt * TILE, dtype=torch.int32, device=dev)
    x0t = pack_vec(xx[:, :, 0], dev)
    p = problem.c_problem(B, x_in_f32=int(f32))
    check(lib().aoc_backward(C.byref(p), int(bool(full_hessian)), _ptr(xt), _ptr(ut), _ptr(x0t), _ptr(Kt),
                             _ptr(lm0), _ptr(st), None, 0), "aoc_backward")
    prm = make_params(stepsize_0=stepsize_0)
    check(lib().aoc_forward(C.byref(p), C.byref(prm), 1, _ptr(xt), _ptr(ut), _ptr(x0t), _ptr(Kt), _ptr(du),
                            _ptr(desc), _ptr(Jn), _ptr(st), None, 0, None), "aoc_forward")
    KK = unpack_gains(Kt, B).cpu().numpy()
    return dict(KK=KK, du=unpack(du, B).cpu().numpy(),
                descent=desc[:B].cpu().numpy(), lmbd0=unpack_vec(lm0, B).cpu().numpy(),
                J_trial0=Jn[:B].cpu().numpy(), status=st[:B].cpu().numpy())


def lqr_tracking_batch(problem, xx_opt, uu_opt, delta):
    """lqr_tracking.lqr_tracking for B trajectories (reference lqr_tracking.py:245-283).
    problem carries the tracking weights QQt,RRt,QQT (reference lqr_tracking.py:324-328).
    xx_opt (B,6,T), uu_opt (B,2,T), delta (6,) or (B,6) -> xx_reg (B,6,T), uu_reg (B,2,T), KK (B,2,6,T)."""
    torch = _torch()
    dev = problem.device
    xx_opt = np.asarray(xx_opt, dtype=np.float64)
    B, T = xx_opt.shape[0], problem.T
    nt = ntiles(B)
    xt, ut = pack(xx_opt, dev), pack(uu_opt, dev)
    x0r = xx_opt[:, :, 0] + np.broadcast_to(np.asarray(delta, dtype=np.float64), (B, 6))  # lqr_tracking.py:265
    x0t = pack_vec(x0r, dev)
    xo0 = pack_vec(xx_opt[:, :, 0], dev)
    Kg = alloc_tiled(B, T, 12, dev)
    xr, ur = alloc_tiled(B, T, 6, dev), alloc_tiled(B, T, 2, dev)
    st = torch.zeros(nt * TILE, dtype=torch.int32, device=dev)
    p = problem.c_problem(B)
    check(lib().aoc_lqr_tracking(C.byref(p), _ptr(xt), _ptr(ut), _ptr(xo0), _ptr(x0t), _ptr(Kg), _ptr(xr), _ptr(ur),
                                 _ptr(st)), "aoc_lqr_tracking")
    KK = unpack(Kg, B).cpu().numpy().reshape(B, 2, 6, T)
    return unpack(xr, B).cpu().numpy(), unpack(ur, B).cpu().numpy(), KK, st[:B].cpu().numpy()


def ltv_lqr_batch(AA, BB, QQ, RR, SS, QQf, x0, qq=None, rr=None, qqf=None, device="cuda:0"):
    """Generic ltv_LQR for nb problems (reference optcon.py:533-771).  Time-major inputs per problem:
    AA (nb,T,6,6), BB (nb,T,6,2), QQ (nb,T,6,6), RR (nb,T,2,2), SS (nb,T,2,6), QQf (nb,6,6), x0 (nb,6),
    affine terms qq (nb,T,6), rr (nb,T,2), qqf (nb,6) or None.  Returns KK (nb,T,2,n), PP (nb,T,n,n),
    xx (nb,T,6), uu (nb,T,2), nreg (nb,), nsing (nb,)."""
    torch = _torch()
    aug = qq is not None or rr is not None or qqf is not None  # optcon.py:614
    n = 7 if aug else 6
    d = lambda a: None if a is None else _dev_f64(a, device)
    A, Bm, Q, R, S, Qf, X0 = (d(a) for a in (AA, BB, QQ, RR, SS, QQf, x0))
    nb, T = A.shape[0], A.shape[1]
    q, r, qf = d(qq), d(rr), d(qqf)
    mk = lambda *s: torch.empty(s, dtype=torch.float64, device=A.device)
    KK, PP, xx, uu = mk(nb, T, 2, n), mk(nb, T, n, n), mk(nb, T, 6), mk(nb, T, 2)
    nreg = torch.zeros(nb, dtype=torch.int32, device=A.device)
    nsing = torch.zeros(nb, dtype=torch.int32, device=A.device)
    st = C.c_void_p(torch.cuda.current_stream(A.device).cuda_stream)
    check(lib().aoc_ltv_lqr(nb, T, int(aug), _ptr(A), _ptr(Bm), _ptr(Q), _ptr(R), _ptr(S), _ptr(Qf), _ptr(X0), _ptr(q),
                            _ptr(r), _ptr(qf), _ptr(KK), _ptr(PP), _ptr(xx), _ptr(uu), _ptr(nreg), _ptr(nsing), st),
          "aoc_ltv_lqr")
    return tuple(t.cpu().numpy() for t in (KK, PP, xx, uu, nreg, nsing))



class NewtonBatchSolverF32:
    """The batched Newton iteration in float32 arithmetic (BASELINE configs[2]): every array float32,
    every operation float32 (`aoc_*_f32`).  Not the parity path — the reference computes in fp64 —
    but the same algorithm; used to measure how much accuracy float32 costs (tests/test_gpu_f32.py).
    Interface: the subset of NewtonBatchSolver needed for fixed-iteration runs."""

    def __init__(self, problem, B, params=None):
        torch = _torch()
        self.problem, self.B, self.T = problem, int(B), problem.T
        self.params = params if params is not None else make_params()
        dev = problem.device
        self.nt = ntiles(self.B)
        self.Bp = self.nt * TILE
        f32 = torch.float32
        self.ref32 = problem.ref.to(f32).contiguous()
        self.xb = [alloc_tiled(B, self.T, 6, dev, zero=True, f32=True) for _ in range(2)]
        self.ub = [alloc_tiled(B, self.T, 2, dev, zero=True, f32=True) for _ in range(2)]
        self.ws = torch.empty((lib().aoc_workspace_bytes_f32(self.B, self.T) + 3) // 4, dtype=f32, device=dev)
        f = lambda: torch.zeros(self.Bp, dtype=f32, device=dev)
        self.J = [f(), f()]
        self.descent, self.stepsize = f(), f()
        self.ntrials = torch.zeros(self.Bp, dtype=torch.int32, device=dev)
        self.status = torch.zeros(self.Bp, dtype=torch.int32, device=dev)
        self.x0 = torch.zeros((self.nt, 6, TILE), dtype=f32, device=dev)
        self.cur = self.jcur = self.kk = 0

    def _p(self):
        p = self.problem.c_problem(self.B, x_in_f32=1, x_out_f32=1)
        p.ref = self.ref32.data_ptr()
        return p

    def set_initial_rollout(self, x0, uu):
        """x0 (B,6), uu (B,2,T): the initial iterate is the open-loop rollout of uu from x0 (config 3:
        warm start from a saved optimum)."""
        torch = _torch()
        dev = self.problem.device
        self.x0.copy_(pack_vec(x0, dev).to(torch.float32))
        ut = pack(uu, dev, f32=True)
        p = self._p()
        check(lib().aoc_rollout_cost_f32(C.byref(p), _ptr(self.x0), _ptr(ut), None, None, _ptr(self.xb[0]),
                                         _ptr(self.ub[0]), _ptr(self.J[0]), _ptr(self.status)), "aoc_rollout_cost_f32")
        self.cur = self.jcur = self.kk = 0

    def set_initial_from_x0(self, x0, kp=5.0, kt=2.5):
        torch = _torch()
        dev = self.problem.device
        self.x0.copy_(pack_vec(x0, dev).to(torch.float32))
        p = self._p()
        check(lib().aoc_initial_trajectory_f32(C.byref(p), float(kp), float(kt), _ptr(self.x0), _ptr(self.xb[0]),
                                               _ptr(self.ub[0])), "aoc_initial_trajectory_f32")
        check(lib().aoc_traj_cost_f32(C.byref(p), _ptr(self.xb[0]), _ptr(self.ub[0]), _ptr(self.x0), _ptr(self.J[0])),
              "aoc_traj_cost_f32")
        self.cur = self.jcur = self.kk = 0

    def iterate(self, kk=None):
        if kk is None:
            kk = self.kk
        p = self._p()
        c, n = self.cur, 1 - self.cur
        jc, jn = self.jcur, 1 - self.jcur
        check(lib().aoc_newton_iterate_f32(C.byref(p), C.byref(self.params), int(kk), _ptr(self.xb[c]), _ptr(self.ub[c]),
                                           _ptr(self.x0), _ptr(self.J[jc]), _ptr(self.ws), self.ws.numel() * 4, _ptr(self.xb[n]),
                                           _ptr(self.ub[n]), _ptr(self.J[jn]), _ptr(self.descent), _ptr(self.stepsize),
                                           _ptr(self.ntrials), _ptr(self.status)), "aoc_newton_iterate_f32")
        self.cur, self.jcur, self.kk = n, jn, kk + 1

    def join(self):
        """(interface shared with TwoStreamNewtonSolver: one stream, nothing to wait for)"""

    def current(self):
        xx = unpack(self.xb[self.cur], self.B)
        xx[:, :, 0] = unpack_vec(self.x0, self.B).to(xx.dtype)
        return xx.cpu().numpy(), unpack(self.ub[self.cur], self.B).cpu().numpy()

    def scalars(self):
        B = self.B
        g = lambda t: t[:B].double().cpu().numpy()
        return dict(cost=g(self.J[1 - self.jcur]), descent=g(self.descent), stepsize=g(self.stepsize),
                    ntrials=self.ntrials[:B].cpu().numpy(), cost_new=g(self.J[self.jcur]),
                    status=self.status[:B].cpu().numpy())
