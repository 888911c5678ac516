"""In-process A/B of aoc_tuning knobs on ONE solver (placement fixed): per-pass ms over iterations 0..K-1 at the headline size.
    python tools/knob_ab.py fw_recompute=0 fw_recompute=1 ..."""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch
from aircraftoptimalcontrol_amd import _lib, batch as aoc, problems


def main():
    B, K = 131072, 12
    settings = [dict(kv.split("=") for kv in a.split(",")) if a != "default" else {} for a in (sys.argv[1:] or ["default"])]
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    s = aoc.NewtonBatchSolver(bp, B, prm)
    for rep in range(3):
        for kn in settings:
            with _lib.tuning(**{k: int(v) for k, v in kn.items()}):
                s.set_initial_from_x0(x0); s.ntrials.zero_()
                evs = [s.iterate_timed(k) for k in range(K)]
                torch.cuda.synchronize()
                ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in evs])
                print("rep %d %-28s passes (bw, fw, search, update) %s  sum %.3f" % (rep, kn or "default", np.round(ms[1:].mean(0), 3), ms[1:].mean(0).sum()), flush=True)


if __name__ == "__main__":
    main()
