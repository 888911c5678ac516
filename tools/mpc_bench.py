"""BASELINE configs[4] at per-GPU size: 1024 receding-horizon instances x N re-solves, T = 500."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from aircraftoptimalcontrol_amd import mpc, problems

def main(B=1024, T=500, steps=200, n_newton=2):
    L = T + steps + 10
    full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
    pr = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
    rh = mpc.RecedingHorizon(pr, problems.tracking_weights(), B, T, n_newton=n_newton,
                             sigma=np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002]))
    rh.start(problems.perturbed_x0(pr, B, seed=1), cold_iters=10)
    rh.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        out = rh.step(fetch=(i == steps - 1))   # nothing is read back between steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("instances %d, T %d, %d re-solves x %d Newton iterations: %.2f s = %.2f ms per receding-horizon step, "
          "%.3g instance-steps/s, final mean cost %.4g, non-finite %d" % (
              B, T, steps, n_newton, dt, dt / steps * 1e3, B * steps / dt, np.nanmean(out["cost"]),
              int((~np.isfinite(out["cost"])).sum())))

if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
