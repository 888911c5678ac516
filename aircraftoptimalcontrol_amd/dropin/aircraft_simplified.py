"""Drop-in for the reference module `aircraft_simplified` on the HIP library.

Same names, arguments, return shapes and dtypes as the reference (file:line cited per method); the
arithmetic runs in libaoc_hip.so on cuda:0.  There is no CPU fallback: without the built library or
without a GPU every compute call raises AocError.

    sys.path.insert(0, ".../aircraftoptimalcontrol_amd/dropin")
    from aircraft_simplified import Dynamics, Cost
"""
import numpy as np

from aircraftoptimalcontrol_amd import batch as _b
from aircraftoptimalcontrol_amd import problems as _pb


class Cost:
    """Cost(QQt, RRt, QQT)  (reference aircraft_simplified.py:16-97)."""

    def __init__(self, QQt, RRt, QQT):
        self.QQt, self.RRt, self.QQT = QQt, RRt, QQT

    def _problem(self):
        return _b.BatchProblem(self.QQt, self.RRt, self.QQT, np.zeros((6, 4)), np.zeros((2, 4)), 1e-3)

    def stagecost(self, xx, uu, xx_ref, uu_ref):
        """-> ll (1,1), lx (6,1), lu (2,1), lxx (6,6), lxu (6,2), lux (2,6), luu (2,2)   (:25-69)"""
        ll, lx, lu, _, _ = _b.cost_batch(self._problem(), np.ravel(xx), np.ravel(uu), np.ravel(xx_ref),
                                         np.ravel(uu_ref))
        ns, ni = np.shape(self.QQt)[0], np.shape(self.RRt)[0]
        return (ll.reshape(1, 1), lx.reshape(-1, 1), lu.reshape(-1, 1), np.array(self.QQt, dtype=float),
                np.zeros((ns, ni)), np.zeros((ni, ns)), np.array(self.RRt, dtype=float))

    def termcost(self, xx, xx_ref):
        """-> llT (1,1), lTx (6,1), lTxx (6,6)   (:71-97)"""
        _, _, _, llT, lTx = _b.cost_batch(self._problem(), np.ravel(xx), np.zeros(2), np.ravel(xx_ref), np.zeros(2))
        return llT.reshape(1, 1), lTx.reshape(-1, 1), self.QQT


class Dynamics:
    """Planar aircraft, state X,Z,V,theta,q,gamma, inputs thrust and pitch moment
    (reference aircraft_simplified.py:101-393)."""

    def __init__(self):
        for k, v in _pb.MODEL.items():   # :108-115
            setattr(self, k, v)
        self.m = 12                      # the reference keeps m as an int
        self.ns, self.ni = 6, 2          # :116-117
        self.dt = 1e-3                   # :118 (drivers overwrite it)
        self.Temp = None
        self.eps_init, self.eps_end, self.speedLimit = 1.5, 0.1, 480
        self.epsilon = self.eps_init

    def _model(self):
        return _b.Model(self.cd0, self.cda, self.cla, float(self.m), self.g, self.S, self.rho, self.J, float(self.dt))

    def step(self, xx, uu, *args):
        """-> xxp (6,) float32, fx (6,6) = A^T, fu (2,6) = B^T, fxx, fuu, fux   (:263-393).
        With a costate argument the second-order tensors are contracted with it ((6,6),(2,2),(2,6));
        without, they are the full (6,6,6), (2,2,6), (2,6,6) tensors."""
        x, u = np.ravel(np.asarray(xx, dtype=np.float64)), np.ravel(np.asarray(uu, dtype=np.float64))
        mdl = self._model()
        if args:
            lm = np.ravel(np.asarray(args[0], dtype=np.float64))
            xp, fx, fu, fxx, fuu, fux = _b.step_batch(mdl, x, u, lm)
            return xp[0].astype(np.float32), fx[0], fu[0], fxx[0], fuu[0], fux[0]
        # uncontracted tensors: contract with the six unit costates in one launch
        X, U = np.repeat(x[None], 6, 0), np.repeat(u[None], 6, 0)
        xp, fx, fu, fxx, fuu, fux = _b.step_batch(mdl, X, U, np.eye(6))
        return (xp[0].astype(np.float32), fx[0], fu[0], np.moveaxis(fxx, 0, -1).copy(),
                np.zeros((2, 2, 6)), np.moveaxis(fux, 0, -1).copy())

    def get_initial_trajectory(self, xx_ref, tt):
        """P-controller initial guess (:126-148) -> xx (6,TT), uu (2,TT).  Evaluated in fp64 with the
        float32 state rounding (the reference's own call runs mostly in float32: agreement ~1e-4)."""
        xx_ref = np.asarray(xx_ref, dtype=np.float64)
        T = np.asarray(tt).shape[0]
        prob = _b.BatchProblem(np.eye(6), np.eye(2), np.eye(6), xx_ref[:, :T], np.zeros((2, T)), float(self.dt),
                               model=self._model())
        s = _b.NewtonBatchSolver(prob, 1)
        s.set_initial_from_x0(xx_ref[:, 0][None])
        xx, uu = s.current()
        return xx[0], uu[0]

    def get_equilibrium(self, x0, tt):
        """Trim point (:152-178).  The reference solves a least-squares problem with SciPy and writes
        the thrust into an int array (truncation to 46); both results are returned as captured
        constants (SciPy 1.15.3) — a one-off set-up value, not part of the accelerated path."""
        xx = np.array(x0, dtype=np.float64).copy()
        xx[2], xx[3], xx[5] = _pb.XXE[2], _pb.XXE[3], _pb.XXE[5]
        self.Temp = np.array(x0, dtype=np.float64).copy()
        return xx, np.array([46, 0, 0, 0])

    def dragForce(self, xx):
        """D and dD/dx (:212-236)."""
        xx = np.ravel(xx)
        al = xx[3] - xx[5]
        D = 0.5 * self.rho * (xx[2] ** 2) * self.S * (self.cd0 + self.cda * al ** 2)
        g = np.zeros((self.ns, 1))
        g[2, 0] = self.rho * xx[2] * self.S * (self.cd0 + self.cda * al ** 2)
        g[3, 0] = self.rho * (xx[2] ** 2) * self.S * self.cda * al
        g[5, 0] = -g[3, 0]
        return D, g

    def liftForce(self, xx):
        """L and dL/dx (:238-261)."""
        xx = np.ravel(xx)
        al = xx[3] - xx[5]
        L = 0.5 * self.rho * (xx[2] ** 2) * self.S * self.cla * al
        g = np.zeros((self.ns, 1))
        g[2, 0] = self.rho * xx[2] * self.S * self.cla * al
        g[3, 0] = 0.5 * self.rho * (xx[2] ** 2) * self.S * self.cla
        g[5, 0] = -g[3, 0]
        return L, g


def tensorCont(P, a):
    """sum_i P[:,:,i] a[i]   (:397-404); host-side helper kept for API completeness."""
    a = np.squeeze(a)
    T = np.zeros(P.shape[:-1])
    for i in range(P.shape[-1]):
        T += P[:, :, i] * a[i]
    return T
