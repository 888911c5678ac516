#!/usr/bin/env python3
"""bench.py — throughput of the batched Newton/LQR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 it launches the N ranks itself (a child `torch.distributed.run`, before anything
here touches a GPU) and relays rank 0's line.  --global-batch G fixes the TOTAL work instead of the per-GPU work
(G / N trajectories per rank, "scaling": "strong"), so that 1-, 2- and 4-GPU lines can be taken at the same 2^20.

Workload (BASELINE.json configs[3], the configuration the headline metric is quoted on): the step-maneuver problem
with tf = 1, dt = 2e-3 (T = 500), fp64, random x0 keyed by the global trajectory index, P-controller initial guess,
131 072 trajectories per GPU (2^20 over the 8 GPUs of a node; weak scaling: per-GPU work is fixed).  One "step" =
one full Newton iteration (backward Riccati pass, forward LQR pass with the first Armijo trials, Armijo
back-tracking, update) of every trajectory of the shard; the timed region runs iterations kk = 0..K-1 from the
initial guess with inputs resident in HBM (`aoc_newton_iterate` per iteration; the shard as two half batches on two
HIP streams that never wait for each other, unless --no-overlap).  No data-path collective; one all-reduce of five scalars (RCCL for
N > 1) closes the timed region.

For N > 1 the line proves what ran: `collective` = {backend, world_seen (= dist.get_world_size()), payload_bytes, us:
the all-reduce by itself, outside the headline region}, `per_rank_ms_per_step` = [min, max] over the ranks' own clocks.
A failed RCCL initialisation is an error exit, never a fallback.  After the timed region the process group is torn
down and rank 0 ALONE produces `cpu_baseline`, `rel_err_vs_oracle` and `secondary` (no rank waits in a collective
while rank 0 spends its seconds on the CPU), for every N.

Prints ONE JSON line (rank 0).  `roofline`/`kernels` come from HIP events recorded on the launch stream between the
passes of the same K iterations run once more on ONE stream right after the timed region (under overlap a kernel's
duration is a property of the interleaving; with --no-overlap the timed region itself carries the events).
`cpu_baseline` is the CPU oracle (a C port of the reference's algorithm, oracle/aoc_oracle.c) timed on this host on a
bounded sample, `rel_err_vs_oracle` the GPU's iterates against the oracle's on that same sample.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# Bytes per trajectory-stage.  algorithmic: fp64 elements of the kernel's own formulation x 8
# (SURVEY 8d's convention); stored: what the kernel really moves (states are kept as float32, §3).
BYTES = {"backward": (176, 152), "forward": (192, 168), "linesearch_update": (96, 72)}
ITERATION_BYTES = 496   # whole Newton iteration (its formulation also carries g = B^T lambda + r)


def provenance():
    """What produced this line: the command line, and fingerprints of bench.py and of the library that was loaded (the
    GPU box gets a snapshot without .git, so a commit hash is not available there)."""
    import hashlib
    from aircraftoptimalcontrol_amd import _lib
    sha = lambda path: hashlib.sha256(open(path, "rb").read()).hexdigest()[:16]
    return {"argv": sys.argv[1:], "bench_py_sha16": sha(os.path.abspath(__file__)), "lib_sha16": sha(_lib.library_path()),
            "aoc_version": _lib.lib().aoc_version().decode(), "abi": int(_lib.lib().aoc_abi_version()),
            "env": {k: v for k, v in os.environ.items() if k.startswith("AOC_")}}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=131072)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong scaling: total trajectories over all ranks (per rank = this / N); 0 = weak scaling with --batch-per-gpu")
    ap.add_argument("--cpu-budget-s", type=float, default=15.0,
                    help="wall seconds of the multi-threaded CPU baseline sample (the 1-thread sample gets a third)")
    ap.add_argument("--horizon", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the small configs reported beside the headline")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one stream; the timed region itself carries the per-pass HIP events")
    ap.add_argument("--placement-candidates", type=int, default=7,
                    help="allocation autotuning before the run: build this many solvers, keep the fastest (batch.best_placed); "
                         "1 = take the buffers as first allocated")
    return ap.parse_args()


def launch_ranks(a):
    """--gpus N > 1 without torchrun: start the N ranks as a child process and relay its output.  Nothing in this
    process has touched a GPU (torch is not even imported), and nothing is exec'ed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def kernel_names(ntiles, full):
    """Names of the kernels behind the passes, as rocprofv3 prints them (diagonal weights, float32 state storage)."""
    from aircraftoptimalcontrol_amd import _lib
    import ctypes as C
    t = _lib.Tuning()
    _lib.lib().aoc_get_tuning(C.byref(t))
    fl = "true, false, true, true" if full else "true, false, false, false"   # <diagonal weights, shared reference, full Hessian, costate>
    wl = t.ls_worklist > 0 or (t.ls_worklist < 0 and ntiles > t.split_tiles)
    small = ntiles <= t.split_tiles
    nspec = _lib.lib().aoc_default_nspec(ntiles * 64, 10)
    lin = t.fw_lin > 0 or (t.fw_lin < 0 and nspec <= 2)
    hcut = (t.bw_hcut if t.bw_hcut >= 0 else (16 if ntiles <= 64 else 8 if (ntiles <= 128 and not full) else 0)) if (not full or t.bw_hcut_full) else 0
    chain = "k_hcut_chain6<true>" if t.hcut_chain6 else "k_track_hcut_chain<true>"
    if not t.hcut_chain6 and hcut >= 4 and (t.hcut_pairs >= 2 or (t.hcut_pairs == 1 and hcut >= 12)):
        chain = "k_hcut_pair<true> (maps composed pairwise), k_track_hcut_chain<true>, k_hcut_odd<true>"
    ngroups = -(-nspec // 3)
    wpe = 1 if (t.fw_wpe1 and ntiles * ngroups <= 256) else 2
    duo_max = 256 if t.fw_duo == 1 else t.fw_duo
    duo = duo_max > 0 and nspec > 3 and ntiles * ((nspec + 1) // 2) <= duo_max
    return {
        "backward": ("phase: " + ("k_bw_hcut_lam<true, false, float> (costate maps), " if full else "") +
                     ("k_bw_hcut_map3<true, false, float, %s> (segment maps, a stage on three wavefronts), %s, k_bw_hcut_gains2<true, false, float, %s> (gains)%s: the horizon in %d segments"
                      % ("true" if full else "false", chain, "true" if full else "false",
                         ", k_backward2<true, false, true, true, float> (lanes with an indefinite M)" if full else "", hcut)
                      if (t.hcut_waves >= 2 or (t.hcut_waves == 1 and ntiles * hcut <= 256)) else
                      "k_bw_hcut<true, false, float, false%s> (segment maps), %s, k_bw_hcut<true, false, float, true%s> (gains)%s: the horizon in %d segments"
                      % (", true" if full else "", chain, ", true" if full else "",
                         ", k_backward2<true, false, true, true, float> (lanes with an indefinite M)" if full else "", hcut)))
                    if hcut >= 2 and ntiles * hcut <= 1024 else
                    ("k_backward5<true, false, float>" if t.bw5 else "k_backward4<true, false, false, false, float>")
                    if (not full and ntiles <= min(t.bw4_tiles, t.split_bw_tiles)) else
                    ("k_backward2<%s, float>" if ntiles <= t.split_bw_tiles else "k_backward<%s, float>") % fl,
        # <diagonal, shared reference, 2 speculated trials, states re-computed (the iterates of the run are rollouts), float32 states>
        "forward": ("k_forward_duo<true, false, float>" if duo else "k_forward_lin<true, false, float, %d>" % wpe if lin
                    else "k_forward_split<true, false, float, %d>" % wpe) if small else
                   "k_forward<true, false, 2, %s, float>" % ("true" if t.fw_recompute else "false"),
        "linesearch_update": "k_ls_final_split<true, false, float>" if small else "k_ls_final<true, false, float>",
        "linesearch_search": ("phase: k_ls_init_wl, k_ls_plan_wl, k_ls_trial_wl<true, false, %d, pinned|plain> x2, k_ls_replan" % max(t.ls_cpl, 1)) if wl
        else "phase: k_ls_init, k_ls_plan, k_ls_trial*, k_ls_resolve (round-based search)",
    }


def oracle_sample(pr, x0, iters, cores, budget, n_fixed=None):
    """The oracle on `cores` threads over the first trajectories of the shard: (n, seconds, final XI, UI, history).
    n is sized for about `budget` seconds of wall time unless n_fixed says how many."""
    from oracle import oracle as orc
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = orc.params(stepsize_0=1.0, armijo_maxiters=10)
    mdl = orc.default_model(pr.dt)
    if n_fixed is None:
        nb = min(4 * cores, x0.shape[0])       # calibrate on 4 trajectories per core
        XI, UI = orc.initial_guess_batch(mdl, pr.xx_ref, x0[:nb], nthreads=cores)
        t0 = time.time()
        orc.newton_iterate_batch(op, prm, XI, UI, XI[:, :, 0].copy(), 0, 1, nthreads=cores)
        rate = nb / max(time.time() - t0, 1e-4)          # trajectory-iterations per second
        n = int(min(x0.shape[0], 65536, max(cores, rate * 2.7 * budget / iters)))   # later iterations search longer
    else:
        n = int(min(n_fixed, x0.shape[0]))
    XI, UI = orc.initial_guess_batch(mdl, pr.xx_ref, x0[:n], nthreads=cores)
    t0 = time.time()
    h = orc.newton_iterate_batch(op, prm, XI, UI, XI[:, :, 0].copy(), 0, iters, nthreads=cores)
    return n, time.time() - t0, XI, UI, h


def _timed(torch, fn, reps=1):
    """best wall time of fn() between two device synchronisations"""
    best = None
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def secondary_configs(torch, batch, problems, T, dev, ncand=3):
    """The other BASELINE configs beside the headline, each at its per-GPU size (pure device time, inputs resident):
    configs[1] 4096 perturbed step-maneuver trajectories (fixed iterations, and solved to convergence on the device);
    configs[2] 65 536 acrobatic trajectories at T = 1000 from the reference's saved optimum, float32 build and fp64 path;
    configs[4] 1024 receding-horizon instances x 200 warm-started re-solves; and the headline workload at the
    reference's native horizon T = 1000."""
    from aircraftoptimalcontrol_amd import mpc
    out = {}
    frac = lambda nbytes, sec: nbytes / sec / 1e9 / HBM_PEAK_GBS
    pr = problems.step_maneuver(tf=1.0, dt=1.0 / T)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt, device=dev)
    prm10 = batch.make_params(stepsize_0=1.0, armijo_maxiters=10)

    # ---- configs[1]: 4096 perturbed trajectories, iterations 0..9
    Bs = 4096
    s = batch.NewtonBatchSolver(bp, Bs, prm10)
    x0 = torch.from_numpy(problems.perturbed_x0(pr, Bs, seed=20260401)).to(dev)

    def fixed10(sv, x0d):
        sv.set_initial_from_x0(x0d)
        sv.ntrials.zero_()
        return _timed(torch, lambda: [sv.iterate(kk) for kk in range(10)])
    best = min(fixed10(s, x0) for _ in range(3))
    out["configs[1]"] = {"workload": "4096 step-maneuver trajectories (perturbed x0), T=%d, fp64, Newton iterations kk=0..9" % T,
                         "ms_per_iteration": best / 10 * 1e3, "trajectory_iterations_per_s": Bs * 10 / best,
                         "iteration_hbm_frac": frac(ITERATION_BYTES * Bs * T * 10, best)}

    # ---- converge mode (SURVEY 8d config 2 "+ a converge-mode run"): aoc_newton_solve, the reference's whole loop with its
    #      stopping rule on the device (optcon.py:415-505), max_iters = 60
    import ctypes as C
    from aircraftoptimalcontrol_amd._lib import lib as _aoc
    conv = {}
    prm60 = batch.make_params(max_iters=60, stepsize_0=1.0, armijo_maxiters=10)
    for Bc in (4096, 65536, 131072):
        if 300.0 * Bc * T > 0.5 * torch.cuda.mem_get_info(dev)[0]:
            continue
        sv = batch.NewtonBatchSolver(bp, Bc, prm60)
        x0c = torch.from_numpy(problems.perturbed_x0(pr, Bc, seed=20260401)).to(dev)
        best_c, r = None, None
        for rep in range(2):
            sv.set_initial_from_x0(x0c)
            r = sv.solve_on_device(sync_every=4, history=False, to_host=False)
            best_c = r["device_seconds"] if best_c is None else min(best_c, r["device_seconds"])
        n_ti = float(r["iters"].sum())
        # where the time goes: one more solve under aoc_solve_trace -> per generation (batch in flight) its iterations and ms
        rows = np.zeros((512, 6))
        _aoc().aoc_solve_trace(rows.ctypes.data_as(C.c_void_p), rows.shape[0])
        sv.set_initial_from_x0(x0c)
        sv.solve_on_device(sync_every=4, history=False, to_host=False)
        rows = rows[:_aoc().aoc_solve_trace_rows()].copy()
        _aoc().aoc_solve_trace(None, 0)
        gens = []
        for part in sorted(set(rows[:, 0].astype(int))):
            rp = rows[(rows[:, 0] == part) & (rows[:, 1] >= 0)]
            dtm = np.diff(np.concatenate([[0.0], rp[:, 5]]))
            for bsz in sorted(set(rp[:, 2].astype(int)), reverse=True):
                m = rp[:, 2] == bsz
                gens.append({"part": int(part), "in_flight": int(bsz), "tiles": int(rp[m, 3][0]), "kk": [int(rp[m, 1].min()), int(rp[m, 1].max())],
                             "ms": round(float(dtm[m].sum()), 3)})
        conv[str(Bc)] = {"device_s": best_c, "solved_trajectories_per_s": Bc / best_c,
                         "iterations_mean": float(r["iters"].mean()), "iterations_max": int(r["iters"].max()),
                         "fraction_converged": float(r["converged"].mean()),
                         "trajectory_iterations_per_s": n_ti / best_c,
                         "iteration_hbm_frac": frac(ITERATION_BYTES * n_ti * T, best_c),
                         "streams": 2 if len(set(rows[:, 0].astype(int))) > 1 else 1,
                         "generations": gens}
        del sv, r
        torch.cuda.empty_cache()
    conv["workload"] = "perturbed step-maneuver trajectories, T=%d, fp64, every trajectory iterated until its descent test stops it " \
                       "(max_iters 60), aoc_newton_solve2: stopping rule, return index, re-packing into denser generations and (from " \
                       "2048 tiles) two halves on two streams, all driven from the C loop; `generations`: batch in flight, iterations kk " \
                       "it ran and their milliseconds (aoc_solve_trace)" % T
    out["converge"] = conv
    torch.cuda.empty_cache()

    # ---- the headline workload at the reference's native horizon (every script of the reference uses T = 1000)
    T2, B2 = 1000, 65536
    pr2 = problems.step_maneuver(tf=1.0, dt=1.0 / T2)
    bp2 = batch.BatchProblem(pr2.QQt, pr2.RRt, pr2.QQT, pr2.xx_ref, pr2.uu_ref, pr2.dt, device=dev)
    x02 = torch.from_numpy(problems.random_x0(B2, seed=20260403)).to(dev)
    s2, pl = batch.best_placed(lambda: batch.NewtonBatchSolver(bp2, B2, prm10), x02, ncand)   # allocation chosen as for the headline
    best = min(fixed10(s2, x02) for _ in range(2))
    out["T1000"] = {"workload": "the headline workload at T=1000 (tf=1, dt=1e-3, main_newton_method.py:71-75): %d trajectories "
                                "from random x0, fp64, Newton iterations kk=0..9, one stream" % B2,
                    "ms_per_iteration": best / 10 * 1e3, "trajectory_iterations_per_s": B2 * 10 / best,
                    "iteration_hbm_frac": frac(ITERATION_BYTES * B2 * T2 * 10, best), "placement_tuning": pl}
    del s2
    torch.cuda.empty_cache()

    # ---- configs[2]: 65 536 acrobatic trajectories, T = 1000, warm start from the reference's saved optimum
    d = np.load(os.path.join(ROOT, "tests", "golden", "data_acrobatic_star.npz"))
    pa = problems.acrobatic()
    bpa = batch.BatchProblem(pa.QQt, pa.RRt, pa.QQT, pa.xx_ref, pa.uu_ref, pa.dt, device=dev)
    Ba, Ta, ITa = 65536, pa.T, 5
    rng = np.random.default_rng(20260402)
    x0a = torch.from_numpy(d["xx_star"][:, 0][None] + rng.normal(0, 1, (Ba, 6)) * problems.SIGMA_X0).to(dev)
    uu_star = d["uu_star"].copy()
    uu_star[:, -1] = 0.0
    uu0 = torch.from_numpy(uu_star).to(dev)[None].expand(Ba, 2, Ta).contiguous()
    s32, pl32 = batch.best_placed(lambda: batch.NewtonBatchSolverF32(bpa, Ba, prm10), x0a, ncand)

    def run32():
        s32.set_initial_rollout(x0a, uu0)
        return _timed(torch, lambda: [s32.iterate(k) for k in range(ITa)])
    t32 = min(run32() for _ in range(2))
    J32 = s32.scalars()["cost_new"][:1024].copy()
    del s32
    torch.cuda.empty_cache()
    s64, pl64 = batch.best_placed(lambda: batch.NewtonBatchSolver(bpa, Ba, prm10), x0a, ncand)

    def run64():
        s64.set_initial_from_rollout(x0a, uu0)
        s64.ntrials.zero_()
        return _timed(torch, lambda: [s64.iterate(k) for k in range(ITa)])
    t64 = min(run64() for _ in range(2))
    J64 = s64.scalars()["cost_new"][:1024].copy()
    del s64, uu0
    torch.cuda.empty_cache()
    ok = np.isfinite(J32) & np.isfinite(J64)
    rel = np.abs(J32 - J64)[ok] / np.abs(J64)[ok]
    out["configs[2]"] = {
        "workload": "65536 acrobatic trajectories (acrobatic_newton.py:34-154), T=%d, warm start = rollout of the reference's "
                    "Data/uu_star_acrobatic.npy from x0 = xx_star[:,0] + N(0, sigma^2), Newton iterations kk=0..%d, one stream" % (Ta, ITa - 1),
        "f32": {"ms_per_iteration": t32 / ITa * 1e3, "trajectory_iterations_per_s": Ba * ITa / t32,
                "iteration_hbm_frac": frac(ITERATION_BYTES // 2 * Ba * Ta * ITa, t32),
                "note": "float32 arithmetic and storage everywhere (aoc_*_f32): 248 B per trajectory-stage", "placement_tuning": pl32},
        "f64": {"ms_per_iteration": t64 / ITa * 1e3, "trajectory_iterations_per_s": Ba * ITa / t64,
                "iteration_hbm_frac": frac(ITERATION_BYTES * Ba * Ta * ITa, t64), "placement_tuning": pl64},
        "cost_rel_f32_vs_f64": {"n": int(ok.sum()), "max": float(rel.max()) if rel.size else None,
                                "median": float(np.median(rel)) if rel.size else None}}

    # ---- configs[4]: 1024 receding-horizon instances (per-GPU share of 8192) x 200 re-solves
    steps, Bm = 200, 1024
    L = T + steps + 10
    full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
    prm_ = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
    rh = mpc.RecedingHorizon(prm_, problems.tracking_weights(), Bm, T, n_newton=2,
                             sigma=np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002]), device=dev)
    rh.start(problems.perturbed_x0(prm_, Bm, seed=1), cold_iters=10)
    rh.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps - 1):
        rh.step(fetch=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    last = rh.step()
    st = rh.solver.status[:Bm].cpu().numpy()
    out["configs[4]"] = {"workload": "1024 receding-horizon instances (per-GPU share of 8192), T=%d, %d re-solves x 2 Newton "
                                     "iterations + tracking gains + plant step, all on the device" % (T, steps),
                         "ms_per_receding_horizon_step": dt / (steps - 1) * 1e3, "instance_steps_per_s": Bm * (steps - 1) / dt,
                         "resolves": steps, "n_nonfinite": int((~np.isfinite(last["cost"])).sum() + (~np.isfinite(last["x_true"])).any(1).sum()),
                         "status_or": int(np.bitwise_or.reduce(st)), "window_pointer": int(rh.s)}
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(pr, x0, iters, budget):
    """cpu_baseline block + what rel_err_vs_oracle needs.  Threads: the CPU share of a one-GPU box is 16 cores.
    The oracle is rebuilt for THIS host first (-O3 -march=native; oracle.use_native() compiles it here and
    accepts it only if it reproduces the portable build bit for bit); if that cannot be done the portable -O2 build is
    timed and the line says so."""
    from oracle import oracle as orc
    vis = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(vis, int(os.environ.get("AOC_CPU_THREADS", "16"))))
    try:
        flags, note = orc.use_native(True), None
    except Exception as e:
        flags, note = orc.PORTABLE_FLAGS, "native build refused: %r" % (e,)
    n, dt, XI, UI, h = oracle_sample(pr, x0, iters, cores, budget)
    n1, dt1, _, _, _ = oracle_sample(pr, x0, iters, 1, budget / 3.0)
    blk = {"value": n * iters / dt, "unit": "trajectory-Newton-iterations/s", "cores": cores, "kind": "port",
           "cpu_model": cpu_model(), "cores_visible": vis, "build_flags": "gcc " + flags,
           "sample": "first %d trajectories x %d iterations of the same workload, oracle/aoc_oracle.c (C port of the "
                     "reference algorithm, fp64) with OpenMP over trajectories, %.1f s" % (n, iters, dt),
           "value_1thread": n1 * iters / dt1,
           "sample_1thread": "first %d trajectories x %d iterations, 1 thread, %.1f s" % (n1, iters, dt1)}
    if note:
        blk["build_note"] = note
    return blk, (n, XI, UI, h), cores


def _compare(batch, bp, prm, x0, sample, iters):
    """GPU (free-running, results do not depend on the batch a trajectory is solved in) against the oracle on the
    same trajectories and iterations kk = 0..iters-1."""
    import torch
    n, XO, UO, h = sample
    s = batch.NewtonBatchSolver(bp, n, prm)
    s.set_initial_from_x0(torch.from_numpy(x0[:n]).to(bp.device))
    J = np.zeros((n, iters)); st = np.zeros((n, iters)); nt = np.zeros((n, iters), np.int32)
    for k in range(iters):
        s.iterate(k)
        sc = s.scalars()
        J[:, k], st[:, k], nt[:, k] = sc["cost"], sc["stepsize"], sc["ntrials"]
    xg, ug = s.current()
    fin_g = np.isfinite(J).all(1) & np.isfinite(ug).all((1, 2))
    fin_o = np.isfinite(h["cost"]).all(1) & np.isfinite(UO).all((1, 2))
    fin = fin_g & fin_o
    same = fin & (st == h["stepsize"]).all(1) & (nt == h["ntrials"]).all(1)      # identical Armijo histories
    with np.errstate(invalid="ignore", divide="ignore"):
        cost_rel = (np.abs(J - h["cost"]) / np.abs(h["cost"])).max(1)
        d = np.abs(ug - UO)
        chan = (d.max(2) / np.maximum(np.abs(UO).max(2), 1e-3)).max(1)             # per input channel: max_t |du| / max_t |u|
        elem = (d / np.maximum(np.abs(UO), 1e-3)).max((1, 2))                      # SURVEY 8c gate: elementwise, floor 1e-3
    xsame = np.array([np.array_equal(xg[b], XO[b]) for b in range(n)])
    g = lambda a, f: float(f(a)) if a.size else None
    q = lambda a: {"max": g(a, np.max), "median": g(a, np.median), "p99.9": g(a, lambda v: np.percentile(v, 99.9))}
    strict = same & xsame
    return {"n": int(n), "iterations": int(iters), "finite": int(fin.sum()),
            "nonfinite_gpu": int((~fin_g).sum()), "nonfinite_oracle": int((~fin_o).sum()),
            "same_nonfinite_set": bool(np.array_equal(fin_g, fin_o)),
            # FIRST the figures with no selection: every trajectory finite on both sides, rounding flips of a float32 state
            # (after which the next iteration's inputs move by ~1e-5) included
            "all_finite": {"n": int(fin.sum()), "cost_rel": q(cost_rel[fin]), "u_rel_channel": q(chan[fin]),
                           "u_rel_elementwise_floor1e-3": q(elem[fin]),
                           "n_u_channel_over_1e-6": int((chan[fin] > 1e-6).sum())},
            "identical_step_and_trial_history": int(same.sum()),
            "states_bit_identical": int(strict.sum()),
            # ... then over the trajectories whose Armijo history and float32 state trajectory equal the oracle's
            "cost_rel_max": g(cost_rel[same], np.max), "cost_rel_median": g(cost_rel[same], np.median),
            "u_rel_channel_max": g(chan[strict], np.max), "u_rel_channel_median": g(chan[strict], np.median),
            "u_rel_elementwise_floor1e-3_max": g(elem[strict], np.max),
            "u_rel_elementwise_floor1e-3_median": g(elem[strict], np.median),
            "u_rel_elementwise_floor1e-3_p99.9": g(elem[strict], lambda v: np.percentile(v, 99.9))}


def teacher_forced_rel_err(batch, bp, prm, pr, x0, n, iters, cores):
    """The ONE scalar for the metric's "fp64 rel-err" half: every iteration kk = 0..iters-1 of the
    first n trajectories of the shard is redone by the oracle FROM THE DEVICE'S OWN ITERATE (teacher-forced: one
    iteration of the reference's algorithm on identical inputs, which is what "matches the reference on identical
    inputs" can mean for a map that is discontinuous in the float32 roundings of its states), and the new inputs are
    compared.  rel_err = max over all trajectory-iterations with identical Armijo verdicts and unregularised gains of
    max_c max_t |u_hip - u_oracle| / max(max_t |u_oracle|, 1e-3) — the size of the input channel is the scale.  Beside
    it SURVEY 8c's elementwise figure (|du| / max(|u|, 1e-3)): median, 99.9th percentile, max, count above 1e-8."""
    import torch
    from oracle import oracle as orc
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    oprm = orc.params(stepsize_0=1.0, armijo_maxiters=10)
    s = batch.NewtonBatchSolver(bp, n, prm)
    s.set_initial_from_x0(torch.from_numpy(x0[:n]).to(bp.device))
    chan_max, elem_all, n_cmp, n_mis, n_reg, cost_max, flips, t_or = 0.0, [], 0, 0, 0, 0.0, 0, 0.0
    per_kk = []
    for kk in range(iters):
        xi, ui = s.current()
        s.iterate(kk)
        sc = s.scalars()
        xn, un = s.current()
        X, U = np.ascontiguousarray(xi), np.ascontiguousarray(ui)
        t0 = time.time()
        r = orc.newton_iterate_batch(op, oprm, X, U, xi[:, :, 0].copy(), kk, 1, nthreads=cores)
        t_or += time.time() - t0
        fin = np.isfinite(sc["cost_new"]) & np.isfinite(r["cost"][:, 0]) & np.isfinite(un).all((1, 2)) & np.isfinite(U).all((1, 2))
        same = fin & (r["stepsize"][:, 0] == sc["stepsize"]) & (r["ntrials"][:, 0] == sc["ntrials"])
        flagged = ((sc["status"] & (4 | 8)) != 0) | (r["nreg"][:, 0] > 0)
        s.status.zero_()
        ok = same & ~flagged
        d = np.where(np.isfinite(un) & np.isfinite(U), np.abs(un - U), 0.0)
        Uf = np.where(np.isfinite(U), U, 0.0)
        chan = (d.max(2) / np.maximum(np.abs(Uf).max(2), 1e-3)).max(1)
        elem = (d / np.maximum(np.abs(Uf), 1e-3)).max((1, 2))
        with np.errstate(invalid="ignore", divide="ignore"):
            crel = np.abs(r["cost"][:, 0] - sc["cost"]) / np.abs(sc["cost"])
        n_cmp += int(ok.sum()); n_mis += int((fin & ~same).sum()); n_reg += int((flagged & fin).sum())
        if ok.any():
            chan_max = max(chan_max, float(chan[ok].max()))
            cost_max = max(cost_max, float(crel[fin].max()))
            elem_all.append(elem[ok])
            flips += int(sum(not np.array_equal(xn[b], X[b]) for b in np.nonzero(ok)[0]))
            per_kk.append(float(chan[ok].max()))
    el = np.concatenate(elem_all) if elem_all else np.zeros(0)
    g = lambda f: float(f(el)) if el.size else None
    return chan_max, {
        "definition": "max over %d teacher-forced trajectory-iterations (first %d trajectories of the shard, kk = 0..%d, each "
                      "redone by the oracle from the device's own iterate; identical Armijo verdicts, gains not regularised) "
                      "of max_c max_t |u_hip - u_oracle| / max(max_t |u_oracle|, 1e-3)" % (n_cmp, n, iters - 1),
        "n": int(n), "iterations": int(iters), "comparable_trajectory_iterations": n_cmp,
        "u_channel_rel_max": chan_max, "u_channel_rel_max_per_iteration": per_kk,
        "u_elementwise_rel_floor1e-3": {"median": g(np.median), "p99.9": g(lambda v: np.percentile(v, 99.9)), "max": g(np.max),
                                        "n_over_1e-8": int((el > 1e-8).sum())},
        "cost_rel_max": cost_max, "armijo_verdict_mismatches": n_mis, "regularised_or_singular_excluded": n_reg,
        "float32_state_rounding_flips": flips, "oracle_seconds": round(t_or, 2),
        "reference_noise_floor": "profiles/r05_oracle_noise_floor.json (the oracle against itself under a 1-ulp change of uu: "
                                 "elementwise p99.9 1.2e-8 / max 1.8e-8 at kk = 0) and profiles/r05_oracle_fma_sensitivity.json (the "
                                 "oracle compiled with fused multiply-adds against itself without: median 6.9e-11, p99.9 3.7e-8, "
                                 "max 5.5e-8 — the level of the device path, whose Riccati algebra uses FMAs)"}


def rel_err_vs_oracle(batch, bp, prm, pr, x0, sample, iters, K, cores):
    """The "fp64 rel-err" half of the metric.  Main block: the cpu_baseline sample (first n trajectories, iterations
    0..iters-1).  `late`: a smaller sample over ALL K iterations of the timed region, so that the full-Hessian regime,
    the trajectories that diverge there (NaN) and the exhaustion storms are compared with the oracle too."""
    out = _compare(batch, bp, prm, x0, sample, iters)
    out["note"] = ("u errors in the main fields are over trajectories whose Armijo history and float32 state trajectory equal "
                   "the oracle's; `all_finite` has no selection: the others differ by one float32 rounding flip of a state, "
                   "after which the next iteration's inputs move by ~1e-5 (DESIGN.md section 2)")
    if K > iters:
        n, dt, XI, UI, h = oracle_sample(pr, x0, K, cores, 0.0, n_fixed=4096)
        late = _compare(batch, bp, prm, x0, (n, XI, UI, h), K)
        late["oracle_seconds"] = dt
        out["late"] = late
    return out


# SQ_INSTS_VALU per wavefront and stage of the large-batch kernels (profiles/r05_sq_counters.txt: rocprofv3 --pmc over the
# one-stream run of this workload; round 3's record has the same counts) and the share of its cycles a wavefront of the
# kernel spends issuing them, two such wavefronts per SIMD
SQ_VALU_PER_WAVE_STAGE = {"k_forward<true, false, 2, true, float>": (624.2, 0.505),
                          "k_backward<true, false, false, false, float>": (546.9, 0.336),
                          "k_backward<true, false, true, true, float>": (593.4, 0.366),
                          "k_ls_final<true, false, float>": (221.2, 0.276)}


def vector_issue(dom, units, T):
    """The dominant kernel against the fp64 issue roof: 1024 SIMDs, one vector instruction of a wavefront per 4 cycles,
    2.4 GHz nominal (the clock held under this load is lower)."""
    for name, (per_stage, issuing) in SQ_VALU_PER_WAVE_STAGE.items():
        if dom["kernel"] == name:
            insts = per_stage * units / 64.0 * (T - 1) / T      # a wavefront walks T - 1 stages (tools/pmc_sq.py divides by them)
            floor_ms = insts / (1024 * 2.4e9 / 4) * 1e3
            return {"valu_insts_per_launch": insts, "issue_ms_at_2.4GHz": floor_ms, "frac": floor_ms / dom["avg_ms"],
                    "wavefront_cycles_issuing_valu": issuing, "wavefronts_per_simd": 2,
                    "source": "profiles/r05_sq_counters.txt (SQ counters of the one-stream run, per wavefront-stage)"}
    return None


def run(a):
    import torch
    import torch.distributed as dist
    from aircraftoptimalcontrol_amd import batch, problems, sharding

    rank, local_rank, world = sharding.env_rank_world()
    # Rehearsal knobs (not used by the driver): AOC_BENCH_BACKEND=gloo runs the collective over gloo on host copies,
    # AOC_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 (several ranks on a one-GPU box).
    backend = os.environ.get("AOC_BENCH_BACKEND", "nccl")
    one_dev = os.environ.get("AOC_BENCH_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if (world == 1 or one_dev) else local_rank)
    if os.environ.get("AOC_BENCH_DEVICE"):   # tests/test_host_logic.py only: the control flow of run() on the CPU, solver stubbed out
        dev = torch.device(os.environ["AOC_BENCH_DEVICE"])
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    sharding.init_process_group(backend, dev)      # raises (non-zero exit) if RCCL cannot initialise: no fallback
    if world > 1:
        assert dist.is_initialized() and dist.get_world_size() == world and dist.get_backend() == backend, \
            "process group: backend %s, world %d; asked for %s, %d" % (dist.get_backend(), dist.get_world_size(), backend, world)

    T, K = a.horizon, a.steps
    strong = a.global_batch > 0
    if strong and a.global_batch % world:
        sys.stderr.write("bench.py: --global-batch %d is not a multiple of %d ranks\n" % (a.global_batch, world))
        sys.exit(2)
    Bg = a.global_batch // world if strong else a.batch_per_gpu
    pr = problems.step_maneuver(tf=1.0, dt=1.0 / T)
    assert pr.T == T
    first, n_own = sharding.shard_range(rank, world, Bg * world)       # contiguous ranges of the global index
    assert n_own == Bg
    x0 = problems.random_x0(Bg, seed=20260403, first=first)            # synthetic inputs of this rank's shard
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt, device=dev)
    prm = batch.make_params(max_iters=200, stepsize_0=1.0, cc=0.5, beta=0.7, armijo_maxiters=10)
    x0d = torch.from_numpy(x0).to(dev)
    # Allocation autotuning, before anything is timed: the duration of the write-heavy passes is a property of the physical
    # pages an allocation happened to get (DRAM write-credit stalls, DESIGN.md section 4 "placement"), stable for its
    # lifetime, and worth 5-10 % between the best and a typical draw: a few solvers are built, each is timed over a few
    # iterations, the fastest is kept (batch.best_placed; the measurements are in `placement_tuning`).
    ncand = max(a.placement_candidates, 1)
    big = batch.ntiles(Bg) >= 1024
    # all candidates are alive at once: never more than fit (a solver holds ~270 B per trajectory-stage: three iterates,
    # K~, du, scratch), and one solver of each kind must remain possible.  Ranks that share a device (one-GPU rehearsal)
    # each budget their share of what is free, so that the code path eight ranks take on eight devices — candidates,
    # fits(), teardown — runs in the rehearsal too.
    per_solver = 270.0 * Bg * T
    share = world if (one_dev and world > 1) else 1
    free0 = torch.cuda.mem_get_info(dev)[0]
    if share > 1:   # every rank looks at the free memory before any of them allocates
        dist.barrier()
    fits = lambda: max(int((0.8 * free0 / share - (free0 - torch.cuda.mem_get_info(dev)[0]) / share) / per_solver), 1) if share > 1 \
        else max(int(0.8 * torch.cuda.mem_get_info(dev)[0] / per_solver), 1)
    placement = {"candidates": ncand if big else 1}
    # test hook (the 2-rank rehearsal): rank r takes candidate r instead of the fastest, so that the ranks' choices DIFFER
    # and every branch that depends on a rank's own choice is taken differently by the ranks
    force = rank if os.environ.get("AOC_BENCH_FORCE_CHOICE", "0") == "1" else None
    # one stream: the attribution pass (per-kernel durations behind `roofline` / `kernels`), or everything with --no-overlap;
    # its allocation is chosen like the headline's (the write-heavy passes differ by 10-20 % between allocations)
    n1 = ncand if big else 1
    s, placement["one_stream_solver"] = batch.best_placed(lambda: batch.NewtonBatchSolver(bp, Bg, prm), x0d, min(n1, fits()),
                                                          keep_first=a.no_overlap, force=force)
    # two half batches on two streams pay while each half is still a large-batch launch (one wavefront per tile kernels)
    overlap = not a.no_overlap and s.nt >= 2048 and fits() >= 1 and torch.cuda.mem_get_info(dev)[0] / share > 1.1 * per_solver
    if world > 1:   # the timed regions hold collectives: every rank must take the same arrangement (free memory is a
        #             per-rank observation — ranks rehearsing on one device see each other's transient allocations)
        flag = torch.tensor([1.0 if overlap else 0.0], dtype=torch.float64, device=dev)
        overlap = bool(sharding.all_reduce(-flag, "max").item() == -1.0)     # min over the ranks
    s2 = None
    if overlap:
        s2, placement["two_stream_solver"] = batch.best_placed(lambda: batch.TwoStreamNewtonSolver(bp, Bg, prm), x0d,
                                                               min(ncand, fits()), keep_first=True, force=force)
    # the solver a caller who does not choose its allocation gets: candidate 0 of the kind that is timed
    sv_first = placement["two_stream_solver" if overlap else "one_stream_solver"].pop("first", None)
    placement["one_stream_solver"].pop("first", None)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def summary(sv):
        """scalar summary of the shard (aoc_summary: one kernel, no torch arithmetic) + the path's only collective
        (RCCL for N > 1)"""
        return sharding.reduce_summary(sv.summary())

    def timed_region(sv, step, n=None):
        sv.set_initial_from_x0(x0d)
        barrier()
        t0 = time.perf_counter()
        evs = [step(k) for k in range(K if n is None else n)]
        summ = summary(sv)
        barrier()
        own = time.perf_counter() - t0
        every = torch.zeros(world, dtype=torch.float64, device=dev)
        every[rank] = own
        every = sharding.all_reduce(every, "sum").cpu().numpy()        # each rank's own clock
        return float(every.max()), every, summ, evs

    # warmup: W iterations from the initial guess (and one summary, so that no lazily loaded code object is first
    # touched inside the timed region), then reset: the timed region is exactly iterations 0..K-1 of the solve
    res = s2 if overlap else s
    warm = [s2, s] if overlap else [s]
    for sv in warm:
        sv.set_initial_from_x0(x0d)
        for k in range(a.warmup):
            sv.iterate(k)
        summary(sv)
    if sv_first is not None and sv_first is not res:   # this rank's first candidate is not the chosen one: warmed up WITHOUT
        sv_first.set_initial_from_x0(x0d)               # the collective (whether a rank gets here depends on its own draw)
        for k in range(a.warmup):
            sv_first.iterate(k)
        sv_first.summary()
    s.iterate_timed(0)
    if overlap:
        el, every, summ, _ = timed_region(s2, lambda k: s2.iterate(k))
        _, _, _, evs = timed_region(s, lambda k: s.iterate_timed(k))      # attribution: same iterations, one stream
    else:
        el, every, summ, evs = timed_region(s, lambda k: s.iterate_timed(k))
    # the same K iterations on the allocation as FIRST made (what a caller of the API gets without best_placed)
    # (timed_region holds collectives: every rank must take the same branch, whatever ITS choice was — a rank whose chosen
    # candidate is its first one times it once more)
    if placement["candidates"] <= 1:
        el_first = el
    else:
        sv1 = res if (sv_first is None or sv_first is res) else sv_first
        el_first = timed_region(sv1, lambda k: sv1.iterate(k))[0]
        del sv1
    del sv_first
    # SURVEY 8d config 4 is quoted on 10 fixed iterations: the same solver over kk = 0..9 (no full-Hessian storms yet)
    el10 = el if K == 10 else timed_region(res, lambda k: res.iterate(k), 10)[0]
    if K != 10:          # leave the K-iteration results in place for what follows
        timed_region(res, lambda k: res.iterate(k))
    # every rank's draw: [chosen probe ms, first-candidate probe ms] per rank (zeros where nothing was probed)
    pk = placement["two_stream_solver" if overlap else "one_stream_solver"]
    mine = torch.zeros(world, 2, dtype=torch.float64, device=dev)
    if pk["ms_per_iteration"]:
        mine[rank, 0], mine[rank, 1] = pk["ms_per_iteration"][pk["chosen"]], pk["ms_per_iteration"][0]
    placement["per_rank_probe_ms_chosen_first"] = sharding.all_reduce(mine, "sum").cpu().numpy().round(3).tolist()
    chose = torch.zeros(world, dtype=torch.float64, device=dev)
    chose[rank] = float(pk["chosen"])
    placement["per_rank_chosen"] = [int(v) for v in sharding.all_reduce(chose, "sum").cpu().numpy()]
    sc = res.scalars()
    coll = None
    if world > 1:   # the path's one collective by itself, outside the headline region
        vec = res.summary()
        ts = []
        for _ in range(12):
            barrier()
            t0 = time.perf_counter()
            sharding.reduce_summary(vec.clone())
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        ts = ts[2:]
        coll = {"backend": dist.get_backend(), "world_seen": int(dist.get_world_size()), "payload_bytes": int(vec.numel() * 8),
                "what": "all_reduce(sum) of %d fp64 scalars on a device tensor, incl. the synchronisation after it" % vec.numel(),
                "us": float(np.mean(ts) * 1e6), "us_min": float(np.min(ts) * 1e6), "repeats": len(ts),
                "device_per_rank": "cuda:0 for every rank (one-GPU rehearsal)" if one_dev else "cuda:LOCAL_RANK"}
    if os.environ.get("AOC_BENCH_DUMP"):   # tests: this rank's per-trajectory results after the K iterations
        full = {} if os.environ.get("AOC_BENCH_DUMP_LIGHT") == "1" else dict(zip(("xx", "uu"), res.current()))   # light: scalars only
        np.savez(os.path.join(os.environ["AOC_BENCH_DUMP"], "rank%d_of_%d.npz" % (rank, world)), first=first,
                 summary=summ.cpu().numpy(), **full, **sc)
    if world > 1:   # nothing below talks to another rank: rank 0 alone finishes the record
        barrier()
        dist.destroy_process_group()
    if rank != 0:
        return

    # per-pass durations from the HIP events (this rank), by Hessian regime: Gauss-Newton for kk <= 8, full after
    passes = batch.NewtonBatchSolver.PASSES
    ms = np.array([[ev[i].elapsed_time(ev[i + 1]) for i in range(len(passes))] for ev in evs])   # [K][pass]
    full = np.arange(K) > prm.hessian_switch
    units = Bg * T                              # trajectory-stages one launch processes
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    tj = json.load(open(tp)) if os.path.exists(tp) else {}
    same_run = tj.get("batch_per_gpu") == Bg and tj.get("T") == T and tj.get("steps") == K
    def entry(pname, names, pick, which):
        avg = float(ms[pick, passes.index(pname)].mean())
        e = {"pass": pname, "kernel": names[pname], "iterations": which, "launches": int(pick.sum()), "avg_ms": avg}
        if pname in BYTES:
            alg, sto = BYTES[pname]
            if pname == "forward" and ", 2, true, float>" in names[pname]:
                sto -= 24          # the nominal states are re-computed, not read (float32: 6 x 4 B)
            e.update(algorithmic_bytes=alg * units, stored_bytes=sto * units,
                     algorithmic_GBps=alg * units / avg / 1e6, stored_GBps=sto * units / avg / 1e6,
                     frac_algorithmic=alg * units / avg / 1e6 / HBM_PEAK_GBS, frac_stored=sto * units / avg / 1e6 / HBM_PEAK_GBS,
                     traffic=tj.get("kernels", {}).get(names[pname], {}).get("hbm_bytes_per_launch") if same_run else None)
        return e

    kernels = []
    if (~full).any():
        kernels.append(entry("backward", kernel_names(s.nt, False), ~full, "Gauss-Newton, kk <= %d" % prm.hessian_switch))
    if full.any():
        kernels.append(entry("backward", kernel_names(s.nt, True), full, "full Hessian, kk > %d" % prm.hessian_switch))
    for pname in passes[1:]:   # the other passes do not change with the Hessian
        kernels.append(entry(pname, kernel_names(s.nt, False), np.ones(K, bool), "all"))
    # the dominant kernel: the one the iterations spend most time in (launches x average; by average alone the entry flipped
    # between the forward pass and the 11-launch full-Hessian backward pass from box to box, 1.79 against 1.77-1.81 ms)
    dom = max((e for e in kernels if "algorithmic_bytes" in e), key=lambda e: e["avg_ms"] * e["launches"])
    fin_b = int(summ[3].item() - summ[4].item())
    out = {
        "metric": "Newton iters/sec (whole node), batched 6-state T=%d trajectories; fp64 rel-err" % T,
        "value": Bg * world * K / el,
        "unit": "trajectory-Newton-iterations/s",
        "n_gpus": world, "steps": K, "warmup": a.warmup,
        "ms_per_step": el / K * 1e3,
        # the same K iterations on the allocation as first made, i.e. without batch.best_placed (DESIGN.md section 4 "placement")
        "ms_per_step_first_allocated": el_first / K * 1e3,
        "value_first_allocated": Bg * world * K / el_first,
        "steps10": {"workload": "the same solver over kk = 0..9 only (SURVEY 8d config 4: \"10 fixed iterations\")",
                    "ms_per_step": el10 / 10 * 1e3, "value": Bg * world * 10 / el10,
                    "iteration_hbm_frac_per_gpu": ITERATION_BYTES * Bg * T * 10 / el10 / 1e9 / HBM_PEAK_GBS},
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: step-maneuver, T=%d (tf=1, dt=%g), fp64, random x0 by global index, "
                               "P-controller initial guess, %d trajectories per GPU (%d over %d GPU%s), Newton iterations "
                               "kk=0..%d from the initial guess (%d Gauss-Newton, %d full-Hessian; fixed-iteration mode: "
                               "nobody is stopped, so from kk~13 a third of the trajectories exhausts every line search)"
                               % (T, pr.dt, Bg, Bg * world, world, "" if world == 1 else "s", K - 1, int((~full).sum()), int(full.sum())),
                   "batch_per_gpu": Bg, "global_batch": Bg * world, "T": T, "parallelism": "batch-sharded x%d" % world,
                   "streams": "two half batches on two HIP streams (batch.TwoStreamNewtonSolver)" if overlap else "one",
                   "armijo": {"stepsize_0": 1.0, "cc": 0.5, "beta": 0.7, "maxiters": 10}},
        "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["algorithmic_GBps"], "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": dom["frac_algorithmic"], "traffic": dom["traffic"],
                     "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "stored_bytes_per_launch": dom["stored_bytes"],
                     "frac_stored": dom["frac_stored"], "avg_launch_ms": dom["avg_ms"], "launches": dom["launches"],
                     "measured_in": ("one-stream repeat of the same %d iterations right after the timed region" % K)
                     if overlap else "the timed region",
                     # the same kernel against the other roof it touches: vector instructions per wavefront-stage from
                     # the SQ counters of the committed one-stream run, priced at the nominal issue rate
                     "vector_issue": vector_issue(dom, units, T)},
        "kernels": kernels,
        "attribution_ms_per_step": float(ms.sum(1).mean()),
        # whole iteration, per GPU: SURVEY 8d's 496 B per trajectory-stage over the wall time of a step
        "iteration_hbm_frac_per_gpu": ITERATION_BYTES * units * K / el / 1e9 / HBM_PEAK_GBS,
        "per_rank_ms_per_step": [float(every.min() / K * 1e3), float(every.max() / K * 1e3)],
        "placement_tuning": placement,
        "collective": coll,
        "last_iter_mean_armijo_trials": float(summ[2].item() / summ[3].item()),
        "final_mean_cost_finite": float(summ[0].item() / max(fin_b, 1)),
        # trajectories whose cost is NaN/Inf: at kk = 9 the reference switches to the full Hessian (optcon.py:443) and a few
        # per cent of the random starts diverge there, in the oracle as on the GPU — on nearly but not exactly the same set
        # (rel_err_vs_oracle.late: e.g. 268 vs 261 of 4096; the regime is ill-conditioned, DESIGN.md section 7, and gated
        # by tests/test_gpu_sweep.py::test_late_regime_free_running).  Throughput without them:
        "n_nonfinite": int(summ[4].item()),
        "value_finite_only": fin_b * K / el,
        "status_or_rank0": int(np.bitwise_or.reduce(sc["status"])),
        "provenance": provenance(),
    }
    del s, s2, res
    torch.cuda.empty_cache()
    if not a.no_cpu_baseline:
        try:
            it = min(K, 10)
            out["cpu_baseline"], sample, cores = cpu_baseline(pr, x0, it, a.cpu_budget_s)
            out["rel_err"], out["rel_err_detail"] = teacher_forced_rel_err(batch, bp, prm, pr, x0, min(sample[0], 16384), it, cores)
            out["rel_err_vs_oracle"] = rel_err_vs_oracle(batch, bp, prm, pr, x0, sample, it, K, cores)
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out.setdefault("cpu_baseline", None)
            out["cpu_baseline_error"] = repr(e)
    else:
        out["cpu_baseline"] = None
    if not a.no_secondary and Bg >= 65536:
        try:
            out["secondary"] = secondary_configs(torch, batch, problems, a.horizon, dev, min(ncand, 3))
        except Exception as e:  # a report beside the headline, never a reason to lose the bench line
            out["secondary_error"] = repr(e)
    print(json.dumps(out), flush=True)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (a.gpus, world))
        sys.exit(2)
    run(a)


if __name__ == "__main__":
    main()
