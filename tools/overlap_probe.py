"""Experiment (EXPERIMENTS.md): two half batches on two HIP streams against one full batch on one stream.
The halves' kernels interleave on the GPU: while one half sits in its latency-bound line-search rounds the
other streams.  Prints ms per iteration of the whole 131 072-trajectory shard for both arrangements."""
import json
import sys
import time

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    n_it = 10
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = problems.random_x0(B, seed=20260403)
    out = {"B": B}
    for parts in (1, 2, 4):
        n = B // parts
        svs = [aoc.NewtonBatchSolver(bp, n, prm) for _ in range(parts)]
        streams = [torch.cuda.Stream() for _ in range(parts)]
        best = None
        for rep in range(3):
            for i, sv in enumerate(svs):
                sv.set_initial_from_x0(x0[i * n:(i + 1) * n])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for kk in range(n_it):
                for sv, st in zip(svs, streams):
                    with torch.cuda.stream(st):
                        sv.iterate(kk)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n_it * 1e3
            best = dt if best is None else min(best, dt)
        out["ms_per_iteration_%d_stream(s)" % parts] = round(best, 3)
        del svs
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
