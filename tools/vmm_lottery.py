"""Placement experiment: the same solver built several times in a row, its big streams in
  torch      torch's caching allocator (what every earlier round measured)
  one        a vmm.Arena with ONE physical handle per stream (workspace, 3 x states, 3 x inputs)
  c<MiB>     a vmm.Arena backed by physical handles of that many MiB (c2, c64, c1024 ...)
and per draw the one-stream pass times (HIP events between the passes, iterations kk = 1..8) and the iteration time.
    python tools/vmm_lottery.py [B=131072] [draws=6] [modes=torch,one,c2,c64,c1024] [min|rec]
Draws of a mode are alive together (as best_placed keeps its candidates), modes run one after the other."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aircraftoptimalcontrol_amd import batch, problems, vmm


def measure(sv, x0d, iters=9, reps=2):
    best = None
    for _ in range(reps):
        sv.set_initial_from_x0(x0d)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        evs = [sv.iterate_timed(k) for k in range(iters)]
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / iters * 1e3
        ms = np.array([[ev[i].elapsed_time(ev[i + 1]) for i in range(4)] for ev in evs])[1:].mean(0)
        rec = [round(float(v), 3) for v in ms] + [round(wall, 3)]
        best = rec if best is None or rec[-1] < best[-1] else best
    return best


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    draws = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    modes = (sys.argv[3] if len(sys.argv) > 3 else "torch,one,c2,c64,c1024").split(",")
    rec = (sys.argv[4] if len(sys.argv) > 4 else "rec") == "rec"
    T = 500
    dev = torch.device("cuda:0")
    pr = problems.step_maneuver(1.0, 1.0 / T)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt, device=dev)
    prm = batch.make_params(stepsize_0=1.0, armijo_maxiters=10)
    x0d = torch.from_numpy(problems.random_x0(B, seed=20260403)).to(dev)
    out = {"B": B, "T": T, "granularity": {"min": vmm.granularity(0, False), "recommended": vmm.granularity(0, True)},
           "columns": ["backward", "forward", "search", "update", "iteration_wall"], "modes": {}}
    print(json.dumps(out["granularity"]), flush=True)
    for mode in modes:
        keep, rows = [], []
        t0 = time.time()
        for d in range(draws):
            if mode == "torch":
                sv = batch.NewtonBatchSolver(bp, B, prm)
            else:
                chunk = None if mode == "one" else int(mode[1:]) << 20
                need = batch.solver_arena_bytes(B, T, gran=max(vmm.granularity(0, rec), 2 << 20), chunk=chunk)
                ar = vmm.Arena(dev, need, chunk_bytes=chunk, recommended=rec)
                sv = batch.NewtonBatchSolver(bp, B, prm, arena=ar)
            keep.append(sv)
            rows.append(measure(sv, x0d))
            print(mode, d, rows[-1], "ptr_ws=%x" % sv.ws.data_ptr(), flush=True)
        out["modes"][mode] = {"draws": rows, "build_s": round(time.time() - t0, 1)}
        del keep, sv
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
