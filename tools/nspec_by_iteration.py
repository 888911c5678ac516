"""How many Armijo candidates should ride along in the forward pass, iteration by iteration?  One solver (placement fixed),
headline workload, one stream: forward + search ms per iteration for nspec = 1, 2, 3, the mean trial count, and what a
policy that picked the best nspec for every iteration would save against the fixed default (2).
    python tools/nspec_by_iteration.py [B] [K]"""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch
from aircraftoptimalcontrol_amd import _lib, batch as aoc, problems


def main(B=131072, K=20):
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    s = aoc.NewtonBatchSolver(bp, B, prm)
    res = {}
    for rep in range(2):
        for ns in (1, 2, 3):
            with _lib.tuning(nspec=ns):
                s.set_initial_from_x0(x0); s.ntrials.zero_()
                evs, tr = [], []
                for k in range(K):
                    evs.append(s.iterate_timed(k))
                    tr.append(s.ntrials[:B].double().mean())
                torch.cuda.synchronize()
                res[ns] = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in evs])
                trials = np.array([float(t) for t in tr])
    print("kk  mean trials | fw + search (+ update) ms at nspec 1 / 2 / 3 | whole iteration")
    for k in range(K):
        print("%2d   %5.2f     | %s | %s" % (k, trials[k], "  ".join("%.2f+%.2f" % (res[n][k, 1], res[n][k, 2]) for n in (1, 2, 3)),
                                          "  ".join("%.3f" % res[n][k].sum() for n in (1, 2, 3))))
    tot = {n: res[n].sum(1) for n in (1, 2, 3)}
    best = np.minimum(np.minimum(tot[1], tot[2]), tot[3])
    print("mean ms per iteration: nspec 1 %.3f, 2 %.3f, 3 %.3f; best per iteration %.3f (%.1f %% below nspec 2)" % (
        tot[1].mean(), tot[2].mean(), tot[3].mean(), best.mean(), 100 * (1 - best.mean() / tot[2].mean())))


if __name__ == "__main__":
    main(*(int(v) for v in sys.argv[1:]))
