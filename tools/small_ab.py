"""Small batches, a knob at a time: ms per Newton iteration (iterations 0..9 from perturbed x0) and per pass for a list
of aoc_tuning settings.   python tools/small_ab.py B1 B2 ... [knob=value,knob=value ...]   (settings after the sizes;
"default" is always measured first)"""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
from aircraftoptimalcontrol_amd import _lib, batch as aoc, problems

SETTINGS = [("default", {})] + [(a, {k: int(v) for k, v in (kv.split("=") for kv in a.split(","))}) for a in sys.argv[1:] if "=" in a]


def main():
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    for B in [int(a) for a in sys.argv[1:] if "=" not in a] or [4096]:
        x0 = torch.from_numpy(problems.perturbed_x0(pr, B, seed=20260401)).cuda()
        for name, kn in SETTINGS * 2:
            with _lib.tuning(**kn):
                s = aoc.NewtonBatchSolver(bp, B, prm)
                best = None
                for rep in range(4):
                    s.set_initial_from_x0(x0); s.ntrials.zero_()
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for k in range(10):
                        s.iterate(k)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                s.set_initial_from_x0(x0); s.ntrials.zero_()
                evs = [s.iterate_timed(k) for k in range(10)]
                torch.cuda.synchronize()
                ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in evs]).mean(0)
                print("B %6d  %-10s  %.3f ms/iteration   passes (bw, fw, search, update) %s  nspec %d" % (
                    B, name, best / 10 * 1e3, np.round(ms, 3), s.n_spec), flush=True)


if __name__ == "__main__":
    main()
