"""Is it WHERE in the device memory the buffers lie?  One process per ballast size: `ballast` GiB are allocated (and kept)
before a two-stream solver is built; ms per iteration over kk = 0..19.   python tools/ballast_probe.py <GiB>"""
import sys
import time

sys.path.insert(0, ".")
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
    B, K = 131072, 20
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    ballast = torch.empty(int(gib * (1 << 30)), dtype=torch.uint8, device="cuda") if gib > 0 else None
    s2 = aoc.TwoStreamNewtonSolver(bp, B, prm)
    out = []
    for rep in range(3):
        s2.set_initial_from_x0(x0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(K):
            s2.iterate(k)
        s2.join()
        torch.cuda.synchronize(); out.append((time.perf_counter() - t0) / K * 1e3)
    print("ballast %5.0f GiB   %.3f ms per iteration (best of 3; all: %s)" % (gib, min(out), [round(v, 3) for v in out]), flush=True)


if __name__ == "__main__":
    main()
