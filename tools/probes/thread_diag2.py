"""Two host threads, each with a solver and a stream of its own: (a) plain iterations, (b) aoc_newton_solve — against the
same work done one after the other."""
import sys, threading
sys.path.insert(0, ".")
import numpy as np, torch
from aircraftoptimalcontrol_amd import batch as aoc, problems
pr = problems.step_maneuver(1.0, 2e-3)
bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
B = 8192
x0 = torch.from_numpy(problems.perturbed_x0(pr, 2 * B, seed=20260401)).cuda()
prm = aoc.make_params(max_iters=60, stepsize_0=1.0, armijo_maxiters=10)
sv = [aoc.NewtonBatchSolver(bp, B, prm) for _ in range(2)]
st = aoc.concurrent_streams("cuda:0", 2)
ref = []
for i in range(2):
    with torch.cuda.stream(st[i]):
        sv[i].set_initial_from_x0(x0[i * B:(i + 1) * B])
        for kk in range(6):
            sv[i].iterate(kk)
        torch.cuda.synchronize()
        ref.append(sv[i].current()[1].copy())
        sv[i].set_initial_from_x0(x0[i * B:(i + 1) * B])
        ref.append(sv[i].solve_on_device(sync_every=4, history=False, to_host=False, two_streams=False)["iters"].copy())
out = {}
def it(i):
    with torch.cuda.stream(st[i]):
        sv[i].set_initial_from_x0(x0[i * B:(i + 1) * B])
        for kk in range(6):
            sv[i].iterate(kk)
        torch.cuda.current_stream().synchronize()
        out["it", i] = sv[i].current()[1].copy()
def so(i):
    with torch.cuda.stream(st[i]):
        sv[i].set_initial_from_x0(x0[i * B:(i + 1) * B])
        out["so", i] = sv[i].solve_on_device(sync_every=4, history=False, to_host=False, two_streams=False)["iters"].copy()
for name, fn in (("iterate", it), ("solve", so)):
    th = [threading.Thread(target=fn, args=(i,)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    key = "it" if name == "iterate" else "so"
    print(name, "in two concurrent threads equals sequential:", [bool(np.array_equal(out[key, i], ref[2 * i + (0 if key == "it" else 1)], equal_nan=True)) for i in range(2)],
          [float(out[key, i].mean()) for i in range(2)] if key == "so" else "")
