"""Converge mode: can the tail of one part hide behind the bulk of another?  The batch as parts on streams of their own, each
solved by aoc_newton_solve (one host thread per part; ctypes releases the GIL), the later parts started after a delay.
    python tools/stagger_probe.py [B=131072]"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = torch.from_numpy(problems.perturbed_x0(pr, B, seed=20260401)).cuda()
    prm = aoc.make_params(max_iters=60, stepsize_0=1.0, armijo_maxiters=10)
    whole = aoc.NewtonBatchSolver(bp, B, prm)
    best = None
    for rep in range(3):
        whole.set_initial_from_x0(x0)
        r = whole.solve_on_device(sync_every=4, history=False, to_host=False)
        best = r["device_seconds"] if best is None else min(best, r["device_seconds"])
    n_it = int(r["iters"].sum())
    print("B %d: aoc_newton_solve2 (two halves): %.1f ms, %.3g trajectory-iterations/s" % (B, best * 1e3, n_it / best), flush=True)
    del whole
    torch.cuda.empty_cache()
    streams = aoc.concurrent_streams("cuda:0", 4)

    def run(parts, delays):
        cuts = np.cumsum([0] + parts)
        solvers = [aoc.NewtonBatchSolver(bp, n, prm) for n in parts]
        res = [None] * len(parts)
        best = None
        for rep in range(3):
            for i, sv in enumerate(solvers):
                with torch.cuda.stream(streams[i]):
                    sv.set_initial_from_x0(x0[cuts[i]:cuts[i + 1]])
            torch.cuda.synchronize()
            t0 = time.perf_counter()

            res[:] = [None] * len(parts)

            def work(i):
                try:
                    if delays[i] > 0:
                        time.sleep(delays[i] * 1e-3)
                    with torch.cuda.stream(streams[i]):
                        res[i] = solvers[i].solve_on_device(sync_every=4, history=False, to_host=False, two_streams=False)
                except Exception as e:   # a thread that dies silently would leave a short wall time behind
                    res[i] = e
            th = [threading.Thread(target=work, args=(i,)) for i in range(len(parts))]
            for t in th: t.start()
            for t in th: t.join()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            for r_ in res:
                if not isinstance(r_, dict):
                    raise RuntimeError("part failed: %r" % (r_,))
            its = sum(int(r_["iters"].sum()) for r_ in res)
            assert its == n_it, (its, n_it)
            best = dt if best is None else min(best, dt)
        print("  parts %s, delays %s ms: %.1f ms wall, %.3g trajectory-iterations/s" % (parts, delays, best * 1e3, its / best), flush=True)

    h = B // 2
    run([h, h], [0, 0])
    q = B // 8
    for d in (30, 50, 70):
        run([3 * q, 3 * q, 2 * q], [0, 0, d])
    for d in (25, 45):
        run([3 * q, 3 * q, q, q], [0, 0, d, 2 * d])
    run([4 * q, 4 * q], [0, 40])
    run([5 * q, 3 * q], [0, 40])


if __name__ == "__main__":
    main()
