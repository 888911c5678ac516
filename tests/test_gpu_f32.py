"""BASELINE configs[2]: the float32-arithmetic build of the kernels on the acrobatic problem, warm
started from the reference's saved optimum (Data/uu_star_acrobatic.npy), against the fp64 path.
float32 is not the parity path (the reference computes in fp64); this test pins what it costs:
costs agree to ~1e-6 relative, inputs to ~1e-3 (gains of ~1e3 amplify the 6e-8 rounding), and the
sweep over tolerances is written to gpurun_out/f32_sweep.json (full-size run: tools/f32_sweep.py)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_f32_tolerance_sweep_acrobatic():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    os.chdir(ROOT)
    import f32_sweep
    r = f32_sweep.run(B=1024, iters=(1, 2, 5))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(dict(B=1024, result=r), open(os.path.join(ROOT, "gpurun_out", "f32_sweep_test.json"), "w"), indent=1)
    for k in (1, 2, 5):
        e = r[k]
        assert e["n_nonfinite"] == 0
        assert e["cost_rel_median"] < 1e-5 and e["cost_rel_max"] < 1e-3, (k, e)
        assert e["uu_rel_median"] < 1e-2, (k, e)
        assert abs(e["mean_cost32"] - e["mean_cost64"]) < 1e-5 * e["mean_cost64"]
    # the solve makes progress in both arithmetic types
    assert r[5]["mean_cost32"] < r[1]["mean_cost32"] and r[5]["mean_cost64"] < r[1]["mean_cost64"]


def test_f32_matches_fp64_on_one_backward_forward():
    """One Newton iteration of the step-maneuver mini-batch in float32: same accepted steps as fp64
    for the clear-cut first iterations, new cost within 1e-5."""
    from conftest import load_golden
    from aircraftoptimalcontrol_amd import batch, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    x0 = problems.perturbed_x0(pr, 200, seed=11)
    prm = batch.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s64 = batch.NewtonBatchSolver(bp, 200, prm); s64.set_initial_from_x0(x0)
    s32 = batch.NewtonBatchSolverF32(bp, 200, prm); s32.set_initial_from_x0(x0)
    s64.iterate(0); s32.iterate(0)
    a, b = s64.scalars(), s32.scalars()
    assert np.allclose(b["cost"], a["cost"], rtol=2e-6) and np.allclose(b["cost_new"], a["cost_new"], rtol=1e-4)
    assert np.allclose(b["descent"], a["descent"], rtol=1e-3)
    assert np.mean(b["ntrials"] == a["ntrials"]) > 0.97


def _acro_warm_start(B, seed=20260402):
    from conftest import load_golden
    from aircraftoptimalcontrol_amd import problems
    d = load_golden("data_acrobatic_star")
    pr = problems.acrobatic()
    rng = np.random.default_rng(seed)
    x0 = d["xx_star"][:, 0][None] + rng.normal(0, 1, (B, 6)) * problems.SIGMA_X0
    uu0 = np.repeat(d["uu_star"][None], B, 0)
    uu0[:, :, -1] = 0.0
    return pr, x0, uu0


def test_f32_iterates_against_the_oracle_acrobatic():
    """configs[2] against the checker: 256 perturbed acrobatic trajectories, T = 1000, warm start; the float32 HIP
    iterates after 1, 2, 5 and 10 iterations against the oracle's fp64 iterates from the same start."""
    from aircraftoptimalcontrol_amd import batch
    from oracle import oracle as orc
    B = 256
    pr, x0, uu0 = _acro_warm_start(B)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = batch.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s32 = batch.NewtonBatchSolverF32(bp, B, prm)
    s32.set_initial_rollout(x0, uu0)
    # the oracle's initial iterate: the open-loop rollout of the saved inputs (get_update with step 0)
    XO = np.zeros((B, 6, pr.T)); UO = np.zeros((B, 2, pr.T))
    for b in range(B):
        XO[b], UO[b] = orc.get_update(op, 0.0, uu0[b], np.zeros_like(uu0[b]), x0[b])
    x32, u32 = s32.current()
    assert np.abs(x32 - XO).max() < 2e-3            # 1000 float32 steps against fp64 + float32 rounding
    oprm = orc.params(stepsize_0=1.0, armijo_maxiters=10)
    done, rep = 0, {}
    for upto in (1, 2, 5, 10):
        h = orc.newton_iterate_batch(op, oprm, XO, UO, x0, done, upto - done)
        for k in range(done, upto):
            s32.iterate(k)
        done = upto
        x32, u32 = s32.current()
        c32 = s32.scalars()["cost_new"]
        co = np.array([orc.traj_cost(op, XO[b], UO[b]) for b in range(B)])
        ok = np.isfinite(c32) & np.isfinite(co)
        ec = np.abs(c32 - co)[ok] / np.abs(co[ok])
        eu = np.max(np.abs(u32 - UO) / np.maximum(np.abs(UO), 1e-3), axis=(1, 2))[ok]
        rep[upto] = dict(n=int(ok.sum()), cost_rel_max=float(ec.max()), cost_rel_median=float(np.median(ec)),
                         u_rel_max=float(eu.max()), u_rel_median=float(np.median(eu)),
                         frac_u_below_1e_2=float(np.mean(eu < 1e-2)))
        assert ok.all()
        assert ec.max() < 1e-4 and np.median(ec) < 1e-5, (upto, rep[upto])
        # inputs: through the Gauss-Newton iterations; the 10th iteration is the first with the full Hessian, where the
        # float32 costate adds its own error (median 1.3e-2 at the full size)
        med, frac = (1e-2, 0.7) if upto <= 5 else (5e-2, 0.3)
        assert np.median(eu) < med and np.mean(eu < 1e-2) > frac, (upto, rep[upto])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "f32_vs_oracle.json"), "w"), indent=1)


def test_f32_full_size_properties_65536():
    """configs[2] at its full size (65 536 acrobatic trajectories, T = 1000, float32 everywhere): what holds at any
    size — every cost finite, the Armijo inequality for every accepted step (evaluated as the kernel does, in float32),
    strict descent directions, the batch's cost going down — plus the time per iteration."""
    import time
    import torch
    from aircraftoptimalcontrol_amd import batch
    B = 65536
    pr, x0, uu0 = _acro_warm_start(B)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = batch.make_params(stepsize_0=1.0, armijo_maxiters=10)
    s = batch.NewtonBatchSolverF32(bp, B, prm)
    s.set_initial_rollout(x0, uu0)
    f32 = np.float32
    means = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(5):
        s.iterate(k)
        sc = {k_: v for k_, v in s.scalars().items()}
        J, Jn, d, a, nt = (sc[k_] for k_ in ("cost", "cost_new", "descent", "stepsize", "ntrials"))
        assert np.isfinite(J).all() and np.isfinite(Jn).all() and np.isfinite(d).all()
        assert (d < 0).all()
        acc = nt < 10
        lhs = Jn.astype(f32)
        rhs = J.astype(f32) + (f32(0.5) * a.astype(f32)) * d.astype(f32)      # optcon.py:268 in float32
        assert (lhs[acc] <= rhs[acc]).all(), int((lhs[acc] > rhs[acc]).sum())
        assert acc.mean() > 0.9
        means.append(float(Jn.mean()))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    assert means[-1] < means[0]
    json.dump(dict(B=B, T=pr.T, mean_cost_after_iteration=means, ms_per_iteration_incl_host_checks=ms),
              open(os.path.join(ROOT, "gpurun_out", "f32_full_size.json"), "w"), indent=1)
