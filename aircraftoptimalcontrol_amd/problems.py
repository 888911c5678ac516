"""Problem definitions of the reference's driver scripts, as plain data (no plotting, no SciPy).

  step_maneuver()  main_newton_method.py:32-142   (configs 1, 2, 4 of BASELINE.json)
  acrobatic()      acrobatic_newton.py:34-154     (config 3)
  tracking_weights()  lqr_tracking.py:324-328     (config 5 gains)

The reference obtains the trim point with scipy.optimize.least_squares and then truncates the
thrust to an integer because its result vector aliases an int array (aircraft_simplified.py:170-174):
uue = [46, 0].  The solver output is a one-off set-up value, captured here as constants
(SciPy 1.15.3) instead of re-running SciPy.

Input generators for the batched configs (perturbed / random x0) are host-side NumPy: they produce the
synthetic start states; the P-controller initial guess itself is rolled out on the device
(`aoc_initial_trajectory`, `NewtonBatchSolver.set_initial_from_x0`).
"""
from dataclasses import dataclass

import numpy as np

# trim point of the reference (xxe) and its integer-truncated input (uue)
XXE = np.array([0.0, 0.0, 9.724826860039666, 0.5888628837019239, 0.0, -0.1625680049882932])
UUE = np.array([46.0, 0.0])

MODEL = dict(cd0=0.1716, cda=2.395, cla=3.256, m=12.0, g=9.81, S=0.61, rho=1.2, J=0.24)


@dataclass
class ProblemData:
    name: str
    QQt: np.ndarray
    RRt: np.ndarray
    QQT: np.ndarray
    xx_ref: np.ndarray  # (6,T)
    uu_ref: np.ndarray  # (2,T)
    tt: np.ndarray
    tf: float
    dt: float

    @property
    def T(self):
        return self.xx_ref.shape[1]


def _stage_weights():
    m, g, J = MODEL["m"], MODEL["g"], MODEL["J"]
    return np.diag([1e-6, m * g * 0.01, 0.5 * m * 0.001, 0.01, 0.5 * J * 0.001, 1e-6]), 1e-6 * np.eye(2)


def _logistic(s):
    return 1.0 / (1.0 + np.exp(-s))


def step_maneuver(tf=1.0, dt=1e-3):
    """Climb of 2.71 m over 16 m along a sigmoid in Z (main_newton_method.py:52-63, :120-142)."""
    Q, R = _stage_weights()
    QT = Q.copy()
    QT[1, 1] *= 20
    QT[3, 3] = QT[1, 1]
    QT[0, 0] = QT[1, 1]
    T = int(tf / dt)
    tt = np.linspace(0, tf, T)
    xf, zf = 16.0, 2.71
    sg = _logistic((tt - tt[-1] / 2) * T)           # slope = number of samples (:113)
    xr = np.zeros((6, T))
    xr[0] = (xf / tf) * tt
    xr[1] = sg * zf
    xr[2] = np.sqrt((sg * (1 - sg) * zf) ** 2 + (xf / tf) ** 2)
    ur = np.repeat(UUE[:, None], T, axis=1)
    return ProblemData("step", Q, R, QT, xr, ur, tt, float(tf), float(dt))


def acrobatic(tf=1.0, dt=1e-3):
    """Up-and-down Z reference, terminal weight x100, xf = 18, uu_ref = [460,-60]
    (acrobatic_newton.py:52-65, :99-154)."""
    Q, R = _stage_weights()
    QT = Q.copy()
    QT[1, 1] *= 100
    QT[3, 3] = QT[1, 1]
    QT[0, 0] = QT[1, 1]
    T = int(tf / dt)
    tt = np.linspace(0, tf, T)
    xf, zf = 18.0, 2.71
    h = T // 2
    slope = T * 0.1
    raw = np.zeros(T)
    raw[:h] = _logistic((tt[:h] - tt[h] / 2) * slope) * zf
    raw[h:] = _logistic((-tt[:h] + tt[h] / 2) * slope) * zf
    zz = np.zeros(T)
    n45 = int(0.45 * T)
    zz[int(0.05 * T):int(0.50 * T)] = raw[:n45]
    zz[int(0.50 * T):int(0.95 * T)] = raw[-n45:]
    xr = np.zeros((6, T))
    xr[0] = (xf / tf) * tt
    xr[1] = zz
    for i in (2, 4, 5):
        xr[i] = XXE[i]
    ur = np.zeros((2, T))
    ur[0] = UUE[0] * 10
    ur[1] = -60.0
    return ProblemData("acrobatic", Q, R, QT, xr, ur, tt, float(tf), float(dt))


def tracking_weights():
    """lqr_tracking.py:324-328: QQt = 0.01 I with [0,0]=[1,1]=10, RRt = 1e-5 I, QQT = QQt."""
    Q = np.eye(6) * 0.01
    Q[0, 0] = Q[1, 1] = 10.0
    return Q, np.eye(2) * 1e-5, Q.copy()


# ------------------------------------------------------------------------------------------------
# synthetic batch inputs
# ------------------------------------------------------------------------------------------------
SIGMA_X0 = np.array([0.5, 0.5, 0.5, 0.05, 0.1, 0.05])


def perturbed_x0(pr, B, seed=20260401):
    """Config 2: x0_b = xx_ref[:,0] + N(0, diag(sigma)^2)."""
    rng = np.random.default_rng(seed)
    return pr.xx_ref[:, 0][None, :] + rng.normal(0, 1, (B, 6)) * SIGMA_X0


def random_x0(B, seed=20260403, first=0):
    """Config 4: uniform X,Z in [-1,1], V in [12,20], theta,gamma in [-0.2,0.2], q in [-0.5,0.5],
    keyed by the GLOBAL trajectory index (Philox counter) so that shards of any size agree."""
    lo = np.array([-1.0, -1.0, 12.0, -0.2, -0.5, -0.2])
    hi = np.array([1.0, 1.0, 20.0, 0.2, 0.5, 0.2])
    out = np.empty((B, 6))
    blk = 4096  # one Philox stream per block of 4096 global indices
    b = 0
    while b < B:
        gidx = first + b
        k, off = divmod(gidx, blk)
        n = min(blk - off, B - b)
        u = np.random.Generator(np.random.Philox(key=seed, counter=[0, 0, 0, k])).random((blk, 6))
        out[b:b + n] = lo + (hi - lo) * u[off:off + n]
        b += n
    return out
