"""The Gauss-Newton backward pass with the horizon cut in S segments (aoc_tuning.bw_hcut, k_bw_hcut) against the
multi-wavefront sequential kernels: ms per Newton iteration over kk = 0..8 (all Gauss-Newton), per-pass times, and the
iterates after those nine iterations compared with the sequential path (Armijo steps, trial counts, inputs relative to
their channel).  Then the receding-horizon step with both cuts.     python tools/hcut_newton_ab.py [B=4096]"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, mpc, problems
from aircraftoptimalcontrol_amd._lib import tuning


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = torch.from_numpy(problems.perturbed_x0(pr, B, seed=20260401)).cuda()
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s = aoc.NewtonBatchSolver(bp, B, prm)
    ref = None
    for S in (0, 4, 8, 12, 16):
        if S * s.nt > 1024:
            continue
        with tuning(bw_hcut=S):
            best = None
            for rep in range(3):
                s.set_initial_from_x0(x0); s.ntrials.zero_()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for kk in range(9):
                    s.iterate(kk)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 9 * 1e3
                best = dt if best is None else min(best, dt)
            xx, uu = s.current(); sc = s.scalars()
            s.set_initial_from_x0(x0); s.ntrials.zero_()
            evs = [s.iterate_timed(kk) for kk in range(9)]
            torch.cuda.synchronize()
            bw = np.mean([e[0].elapsed_time(e[1]) for e in evs[1:]])
        if ref is None:
            ref = (xx, uu, sc); msg = ""
        else:
            chan = (np.abs(uu - ref[1]).max(2) / np.maximum(np.abs(ref[1]).max(2), 1e-3)).max()
            msg = "  vs sequential: steps equal %s, trials equal %s, u (channel) %.1e, states identical %d / %d" % (
                np.array_equal(sc["stepsize"], ref[2]["stepsize"]), np.array_equal(sc["ntrials"], ref[2]["ntrials"]), chan,
                sum(np.array_equal(xx[b], ref[0][b]) for b in range(B)), B)
        print("B %d bw_hcut %2d: %.3f ms per iteration (kk 0..8), backward pass %.3f ms%s" % (B, S, best, bw, msg), flush=True)
    if B <= 1024:
        T, steps = 500, 200
        L = T + steps + 10
        full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
        prm_ = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
        for St, Sb in ((0, 0), (16, 0), (16, 8), (16, 16), (24, 24), (16, 32)):
            with tuning(track_hcut=St, bw_hcut=Sb):
                rh = mpc.RecedingHorizon(prm_, problems.tracking_weights(), B, T, n_newton=2,
                                         sigma=np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002]))
                rh.start(problems.perturbed_x0(prm_, B, seed=1), cold_iters=10)
                rh.step(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(100):
                    rh.step(fetch=False)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 10
                out = rh.step()
            print("receding horizon, %d instances: track_hcut %2d bw_hcut %2d: %.3f ms per step, mean cost %.6f" % (B, St, Sb, ms, out["cost"].mean()), flush=True)


if __name__ == "__main__":
    main()
