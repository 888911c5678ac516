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
