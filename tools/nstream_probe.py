"""The fixed-iteration headline as N parts on N streams that are checked to run side by side (round 3 measured 1 / 2 / 4
streams = 5.17 / 4.85 / 6.00 ms — with streams that may have shared a hardware queue).   python tools/nstream_probe.py"""
import sys
import time

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems
from aircraftoptimalcontrol_amd._lib import tuning


def main():
    B, K = 131072, 20
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    nt = B // 64
    for N, knobs in ((1, {}), (2, {}), (3, {}), (4, {}), (4, dict(split_tiles=256, split_bw_tiles=256)), (2, {})):
        cuts = [64 * (nt * i // N) for i in range(N + 1)]
        with tuning(**knobs):
            parts = [aoc.NewtonBatchSolver(bp, cuts[i + 1] - cuts[i], prm) for i in range(N)]
            streams = aoc.concurrent_streams("cuda:0", N) if N > 1 else [torch.cuda.current_stream()]
            res = []
            for Kk in (10, K):
                best = None
                for rep in range(3):
                    for i, sv in enumerate(parts):
                        with torch.cuda.stream(streams[i]):
                            sv.set_initial_from_x0(x0[cuts[i]:cuts[i + 1]])
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for k in range(Kk):
                        for i, sv in enumerate(parts):
                            with torch.cuda.stream(streams[i]):
                                sv.iterate(k)
                    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / Kk * 1e3
                    best = dt if best is None else min(best, dt)
                res.append(round(best, 3))
        print("%d part(s) %s: %.3f ms per iteration over kk 0..9, %.3f over kk 0..19" % (N, knobs or "", res[0], res[1]), flush=True)
        del parts
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
