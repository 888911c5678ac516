"""Which lanes does the full-Hessian horizon cut hand to the sequential kernel, and why?  (diagnostic; needs a GPU)
    python tools/hcut_full_stats.py [B=4096] [dist=perturbed|random] [kk_max=14]
Runs Newton iterations kk = 0..kk_max; before every full-Hessian iteration calls aoc_backward on the current iterate with a
scratch region of its own and reads the cut's per-lane flag words back (csrc/passes/hcut.inc: HcutFull.lane_flags)."""
import ctypes as C
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from aircraftoptimalcontrol_amd import batch, problems
from aircraftoptimalcontrol_amd._lib import lib, check

def main(B=4096, dist="perturbed", kk_max=14):
    B, kk_max = int(B), int(kk_max)
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.perturbed_x0(pr, B, seed=20260401) if dist == "perturbed" else problems.random_x0(B, seed=20260403)
    s = batch.NewtonBatchSolver(bp, B, batch.make_params(stepsize_0=1.0, armijo_maxiters=10))
    s.set_initial_from_x0(x0)
    nt, T, S = s.nt, s.T, 16
    nbytes = int(lib().aoc_backward_scratch_bytes(B, T))
    scratch = torch.zeros(nbytes // 8 + 8, dtype=torch.float64, device="cuda")
    Kt = torch.zeros(nt * T * 14 * 64, dtype=torch.float64, device="cuda")
    st = torch.zeros(nt * 64, dtype=torch.int32, device="cuda")
    off_flags = nt * S * (90 + 27) * 64 * 8 + nt * S * (42 + 6) * 64 * 8      # hcut_scratch_bytes + costate maps + lambda
    for kk in range(kk_max + 1):
        if kk > 8:
            p = s._p(1)
            check(lib().aoc_backward(C.byref(p), 1, batch._ptr(s.xb[s.cur]), batch._ptr(s.ub[s.cur]), batch._ptr(s.x0), batch._ptr(Kt),
                                     None, batch._ptr(st), batch._ptr(scratch), nbytes), "aoc_backward")
            torch.cuda.synchronize()
            raw = scratch.view(torch.int32)[off_flags // 4: off_flags // 4 + nt * 64].cpu().numpy()[:B]
            d = (raw & (1 << 30)) != 0
            tiles = len(np.unique(np.nonzero(d)[0] // 64))
            print("kk %2d: distrusted lanes %5d of %d (map %5d, true recursion %5d), tiles with one %3d of %d; regularised %d singular %d"
                  % (kk, d.sum(), B, ((raw & (1 << 29)) != 0).sum(), ((raw & (1 << 28)) != 0).sum(), tiles, nt,
                     ((raw & 8) != 0).sum(), ((raw & 4) != 0).sum()), flush=True)
        s.iterate(kk)

if __name__ == "__main__":
    main(*sys.argv[1:])
