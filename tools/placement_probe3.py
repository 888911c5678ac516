"""Placement sensitivity inside ONE allocation: the K~/du workspace as a view into a big arena at different byte
offsets; backward / forward / update times per offset.  If the time varies with the offset inside the same physical
allocation, the effect is an address-bit one (and can be engineered away); if only across allocations, it is the
physical page set."""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch
from aircraftoptimalcontrol_amd import batch as aoc, problems


def run(s, x0):
    s.set_initial_from_x0(x0)
    evs = [s.iterate_timed(k) for k in range(5)]
    torch.cuda.synchronize()
    ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in evs])
    return np.round(ms[1:].mean(0), 3)


def main():
    B = 131072
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0 = torch.from_numpy(problems.random_x0(B, seed=20260403)).cuda()
    s = aoc.NewtonBatchSolver(bp, B, prm)
    n = s.ws.numel()
    print("base", run(s, x0), flush=True)
    for rep in range(2):
        arena = torch.empty(n + (3 << 27), dtype=torch.float64, device="cuda")     # + 3 GiB of slack
        print("arena %d at %#x" % (rep, arena.data_ptr()), flush=True)
        for off in (0, 512, 4096, 65536, 1 << 20, 2 << 20, 6 << 20, 32 << 20, 256 << 20, 1 << 30, (1 << 30) + (2 << 20), 2 << 30, (3 << 30) - (2 << 20)):
            s.ws = arena[off // 8: off // 8 + n]
            print("  offset %11d B (%#x)  %s" % (off, off, run(s, x0)), flush=True)
        del arena
        s.ws = torch.empty(n, dtype=torch.float64, device="cuda")


if __name__ == "__main__":
    main()
