"""Which buffer's physical placement decides the duration of a pass?  One solver; then, one buffer class at a time, six
alternative allocations are swapped in (all kept alive) and iterations 0..5 are timed pass by pass."""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def run(s, x0):
    s.set_initial_from_x0(x0)
    evs = [s.iterate_timed(k) for k in range(6)]
    torch.cuda.synchronize()
    ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in evs])
    return np.round(ms[1:].mean(0) * 500.0 / s.T, 3)    # per 500 stages


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    pr = problems.step_maneuver(1.0, 1.0 / T)
    assert pr.T == T, (pr.T, T)
    print("T = %d: tile strides K~ %d B (mod 4096 = %d), x %d B (mod 4096 = %d), u %d B (mod 4096 = %d)" % (
        T, T * 7168, T * 7168 % 4096, T * 1536, T * 1536 % 4096, T * 1024, T * 1024 % 4096))
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    s = aoc.NewtonBatchSolver(bp, B, prm)
    print("base                         ", run(s, x0), run(s, x0), flush=True)
    keep = []
    for i in range(6):
        ws = torch.empty_like(s.ws); keep.append(ws)
        s.ws = ws
        print("ws candidate %d  %#x    " % (i, ws.data_ptr()), run(s, x0), flush=True)
    if len(sys.argv) > 3:
        return
    for i in range(6):
        xb = [torch.zeros_like(t) for t in s.xb]; keep.append(xb)
        s.xb = xb
        print("xb candidate %d  %#x    " % (i, xb[0].data_ptr()), run(s, x0), flush=True)
    for i in range(6):
        ub = [torch.zeros_like(t) for t in s.ub]; keep.append(ub)
        s.ub = ub
        print("ub candidate %d  %#x    " % (i, ub[0].data_ptr()), run(s, x0), flush=True)


if __name__ == "__main__":
    main()
