"""A/B of the scheduling knobs in the regime between the two kernel families (513-2048 tiles: one wavefront per SIMD or
fewer with the one-wavefront-per-tile kernels): BASELINE configs[2] (65 536 acrobatic trajectories, T = 1000, warm start
from the reference's saved optimum), float32 build and fp64 path, ms per Newton iteration (kk = 0..4) per setting.
    python tools/tune_mid.py [B=65536] [which=both|f32|f64]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from aircraftoptimalcontrol_amd import _lib, batch, problems

SETTINGS = [
    ("default", {}),
    ("split_bw 1024", dict(split_bw_tiles=1024)),
    ("split_bw 1024 + bw4 1024", dict(split_bw_tiles=1024, bw4_tiles=1024)),
    ("split 1024 (fw/final)", dict(split_tiles=1024)),
    ("split 1024 both", dict(split_tiles=1024, split_bw_tiles=1024)),
    ("split 1024 both + bw4 1024", dict(split_tiles=1024, split_bw_tiles=1024, bw4_tiles=1024)),
    ("nspec 3", dict(nspec=3)),
    ("nspec 1", dict(nspec=1)),
    ("split 1024 both + nspec 3", dict(split_tiles=1024, split_bw_tiles=1024, nspec=3)),
    ("worklist always", dict(ls_worklist=1)),
    ("worklist never", dict(ls_worklist=0)),
]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    which = sys.argv[2] if len(sys.argv) > 2 else "both"
    dev = torch.device("cuda:0")
    d = np.load(os.path.join(ROOT, "tests", "golden", "data_acrobatic_star.npz"))
    pa = problems.acrobatic()
    bp = batch.BatchProblem(pa.QQt, pa.RRt, pa.QQT, pa.xx_ref, pa.uu_ref, pa.dt, device=dev)
    rng = np.random.default_rng(20260402)
    x0 = torch.from_numpy(d["xx_star"][:, 0][None] + rng.normal(0, 1, (B, 6)) * problems.SIGMA_X0).to(dev)
    uu = d["uu_star"].copy(); uu[:, -1] = 0.0
    uu0 = torch.from_numpy(uu).to(dev)[None].expand(B, 2, pa.T).contiguous()
    prm = batch.make_params(stepsize_0=1.0, armijo_maxiters=10)
    IT = 5
    out = {"B": B, "T": pa.T, "iterations": IT, "settings": {}}
    ref = {}
    # ONE solver per arithmetic type for every setting: the duration of a pass depends on where its buffers were allocated
    # ("placement"), so a fresh solver per setting would compare allocations, not settings
    s32 = batch.NewtonBatchSolverF32(bp, B, prm) if which in ("both", "f32") else None
    s64 = batch.NewtonBatchSolver(bp, B, prm) if which in ("both", "f64") else None
    for name, kn in SETTINGS + SETTINGS[:1]:
        row = {}
        with _lib.tuning(**kn):
            if s32 is not None:
                s = s32
                best = None
                for rep in range(3):
                    s.set_initial_rollout(x0, uu0)
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for k in range(IT):
                        s.iterate(k)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                row["f32_ms"] = round(best / IT * 1e3, 3)
                J = s.scalars()["cost_new"]
                row["f32_same"] = bool(np.array_equal(J, ref.setdefault("f32", J), equal_nan=True))
            if s64 is not None:
                s = s64
                best = None
                for rep in range(3):
                    s.set_initial_from_rollout(x0, uu0); s.ntrials.zero_()
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    for k in range(IT):
                        s.iterate(k)
                    torch.cuda.synchronize(); dt = time.perf_counter() - t0
                    best = dt if best is None else min(best, dt)
                row["f64_ms"] = round(best / IT * 1e3, 3)
                J = s.scalars()["cost_new"]
                row["f64_same"] = bool(np.array_equal(J, ref.setdefault("f64", J), equal_nan=True))
                s.set_initial_from_rollout(x0, uu0); s.ntrials.zero_()
                evs = [s.iterate_timed(k) for k in range(IT)]
                torch.cuda.synchronize()
                ms = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(4)] for e in evs]).mean(0)
                row["f64_passes_ms"] = [round(float(v), 3) for v in ms]
        out["settings"].setdefault(name, row)
        print(name, row, flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
