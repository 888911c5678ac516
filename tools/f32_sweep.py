"""BASELINE configs[2]: acrobatic problem, T=1000, warm start from the reference's saved optimum,
float32 arithmetic vs the fp64 path: relative error of (uu, cost) after 1, 2, 5, 10 iterations."""
import json, sys
sys.path.insert(0, '.')
import numpy as np
from aircraftoptimalcontrol_amd import batch, problems

def run(B=65536, seed=20260402, iters=(1, 2, 5, 10)):
    d = np.load('tests/golden/data_acrobatic_star.npz')
    pr = problems.acrobatic()
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    rng = np.random.default_rng(seed)
    x0 = d['xx_star'][:, 0][None] + rng.normal(0, 1, (B, 6)) * problems.SIGMA_X0
    uu0 = np.repeat(d['uu_star'][None], B, 0)
    uu0[:, :, -1] = 0.0
    prm = batch.make_params(stepsize_0=1.0, armijo_maxiters=10)
    # fp64: initial iterate = open-loop rollout
    XI, UI, J0, _ = batch.rollout_cost(bp, x0, uu0)
    s64 = batch.NewtonBatchSolver(bp, B, prm); s64.set_initial(XI, UI)
    s32 = batch.NewtonBatchSolverF32(bp, B, prm); s32.set_initial_rollout(x0, uu0)
    import torch, time
    out = {}
    # throughput of the two arithmetic types on this workload (10 iterations each, then reset)
    for name, sv in (("f64", s64), ("f32", s32)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for k in range(10):
            sv.iterate(k)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        out["ms_per_iteration_" + name] = dt / 10 * 1e3
    s64.set_initial(XI, UI); s32.set_initial_rollout(x0, uu0)
    for k in range(max(iters)):
        s64.iterate(k); s32.iterate(k)
        if k + 1 in iters:
            x64, u64 = s64.current(); x32, u32 = s32.current()
            c64, c32 = s64.scalars()['cost_new'], s32.scalars()['cost_new']
            ok = np.isfinite(c64) & np.isfinite(c32)
            eu = np.max(np.abs(u32 - u64) / np.maximum(np.abs(u64), 1e-3), axis=(1, 2))
            ec = np.abs(c32 - c64) / np.abs(c64)
            same_step = (s64.scalars()['stepsize'] == s32.scalars()['stepsize'].astype(np.float32).astype(np.float64))
            tol = next((t for t in (1e-6, 1e-5, 1e-4, 1e-3, 1e-2) if np.all(eu[ok] < t) and np.all(ec[ok] < t)), None)
            out[k + 1] = dict(uu_rel_max=float(eu[ok].max()), uu_rel_median=float(np.median(eu[ok])),
                              cost_rel_max=float(ec[ok].max()), cost_rel_median=float(np.median(ec[ok])),
                              n_nonfinite=int((~ok).sum()), tightest_tol_all_pass=tol,
                              frac_uu_below={str(t): float(np.mean(eu[ok] < t)) for t in (1e-2, 1e-3, 1e-4, 1e-5, 1e-6)},
                              mean_cost64=float(np.mean(c64[ok])), mean_cost32=float(np.mean(c32[ok])))
    return out

if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    r = run(B)
    print(json.dumps(r, indent=1))
    json.dump(dict(B=B, result=r), open('gpurun_out/f32_sweep.json', 'w'), indent=1)
