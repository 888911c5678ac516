"""Drop-in for the reference module `lqr_tracking` on the HIP library.

    from lqr_tracking import lqr_tracking, ltv_LQR

The reference's lqr_tracking() reads the module globals dyn, ns, ni, QQt, RRt, QQT that only exist
when the file runs as a script (lqr_tracking.py:321-328); here they are module attributes initialised
to those same values, so the function is callable after a plain import as well."""
import numpy as np

from aircraftoptimalcontrol_amd import batch as _b
from aircraftoptimalcontrol_amd import problems as _pb
from aircraft_simplified import Dynamics
from optcon import ltv_LQR, _model_of  # noqa: F401  (the reference module carries its own identical ltv_LQR)

dyn = Dynamics()
ns, ni = dyn.ns, dyn.ni
QQt, RRt, QQT = _pb.tracking_weights()   # lqr_tracking.py:324-328


def lqr_tracking(xx_opt, uu_opt, tt):
    """-> xx_reg (ns,TT), uu_reg (ni,TT)   (lqr_tracking.py:245-283): LQR gains about the optimal
    trajectory and the closed-loop rollout from xx_opt[:,0] + 0.1."""
    xx_opt = np.asarray(xx_opt, dtype=np.float64)
    uu_opt = np.asarray(uu_opt, dtype=np.float64)
    TT = np.asarray(tt).shape[0]
    prob = _b.BatchProblem(QQt, RRt, QQT, np.zeros((6, TT)), np.zeros((2, TT)), float(dyn.dt), model=_model_of(dyn))
    delta_xx = np.ones((6,)) * 0.1      # :259
    xr, ur, _, _ = _b.lqr_tracking_batch(prob, xx_opt[None, :, :TT], uu_opt[None, :, :TT], delta_xx)
    return xr[0], ur[0]


if __name__ == '__main__':
    import os
    import sys
    data = sys.argv[1] if len(sys.argv) > 1 else 'Data'
    xx_opt, uu_opt = np.load(os.path.join(data, 'xx_star.npy')), np.load(os.path.join(data, 'uu_star.npy'))
    tt = np.linspace(0, 1, xx_opt.shape[1])
    xx_lqr, uu_lqr = lqr_tracking(xx_opt, uu_opt, tt)
    print('tracking error at T: %s' % (xx_lqr[:, -1] - xx_opt[:, -1]))
