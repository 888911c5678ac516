"""BASELINE configs[2] timing: 65 536 acrobatic trajectories, T = 1000, float32 arithmetic; ms per Newton iteration
(10 iterations from the warm start, after one untimed pass).  Kernel selection knobs come from the environment (AOC_*)."""
import json, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from aircraftoptimalcontrol_amd import batch, problems

def main(B=65536):
    d = np.load('tests/golden/data_acrobatic_star.npz')
    pr = problems.acrobatic()
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    rng = np.random.default_rng(20260402)
    x0 = d['xx_star'][:, 0][None] + rng.normal(0, 1, (B, 6)) * problems.SIGMA_X0
    uu0 = np.repeat(d['uu_star'][None], B, 0)
    uu0[:, :, -1] = 0.0
    s = batch.NewtonBatchSolverF32(bp, B, batch.make_params(stepsize_0=1.0, armijo_maxiters=10))
    out = {"B": B, "T": pr.T}
    for rep in range(2):
        s.set_initial_rollout(x0, uu0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(10):
            s.iterate(k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out["ms_per_iteration"] = round(dt / 10 * 1e3, 3)
    out["hbm_frac_algorithmic_248B"] = round(B * pr.T * 248 / (out["ms_per_iteration"] * 1e-3) / 8e12, 3)
    print(json.dumps(out))

if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
