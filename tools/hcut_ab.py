"""The tracking gains with the horizon cut in S segments (aoc_tuning.track_hcut, k_track_hcut_*) against the sequential
kernels: (i) gains of one receding-horizon step, max difference relative to the largest gain of the column; (ii) time of
the gains launch by itself (HIP events around aoc_lqr_tracking-equivalent inside aoc_mpc_step is not separable, so the
whole step is timed: n_newton = 0 leaves gains + plant step + warm start) and of the whole step with two iterations.
    python tools/hcut_ab.py [instances=1024]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import mpc, problems
from aircraftoptimalcontrol_amd._lib import tuning


def build(B, T, n_newton):
    steps = 200
    L = T + steps + 10
    full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
    pr = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
    rh = mpc.RecedingHorizon(pr, problems.tracking_weights(), B, T, n_newton=n_newton,
                             sigma=np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002]))
    rh.start(problems.perturbed_x0(pr, B, seed=1), cold_iters=10)
    return rh


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    T = 500
    ref = None
    for S in (0, 4, 8, 12, 16, 24, 32):
        with tuning(track_hcut=S):
            rh = build(B, T, 2)
            rh.step(); torch.cuda.synchronize()
            K = rh.Kg.cpu().numpy().copy()        # tiled [ntiles][T][12][64]: gains of the second step's tracking problem
            out = rh.step()
            t0 = time.perf_counter()
            for i in range(100):
                rh.step(fetch=False)
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t0) * 10
            rh0 = build(B, T, 0)
            rh0.step(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(100):
                rh0.step(fetch=False)
            torch.cuda.synchronize()
            ms0 = (time.perf_counter() - t0) * 10
        if ref is None:
            ref = (K, out)
            err = xerr = 0.0
        else:
            scale = np.abs(ref[0]).max(axis=(1, 3), keepdims=True)
            err = float((np.abs(K - ref[0]) / np.maximum(scale, 1e-300)).max())
            xerr = float(np.abs(out["x_true"] - ref[1]["x_true"]).max())
        print("track_hcut %2d: step with 2 iterations %.3f ms, gains + plant + warm start only %.3f ms; gains vs sequential "
              "(rel. to column scale) %.2e, plant state after 2 steps differs by %.1e" % (S, ms2, ms0, err, xerr), flush=True)


if __name__ == "__main__":
    main()
