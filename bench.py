#!/usr/bin/env python3
"""bench.py — throughput of the batched Newton/LQR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 it launches the N ranks itself (a child `torch.distributed.run`, before anything
here touches a GPU) and relays rank 0's line.

Workload (BASELINE.json configs[3], the configuration the headline metric is quoted on): the step-maneuver problem
with tf = 1, dt = 2e-3 (T = 500), fp64, random x0 keyed by the global trajectory index, P-controller initial guess,
131 072 trajectories per GPU (2^20 over the 8 GPUs of a node; weak scaling: per-GPU work is fixed).  One "step" =
one full Newton iteration (backward Riccati pass, forward LQR pass with the first Armijo trials, Armijo
back-tracking, update) of every trajectory of the shard; the timed region runs iterations kk = 0..K-1 from the
initial guess with inputs resident in HBM (`aoc_newton_iterate` per iteration; the shard as two half batches on two
HIP streams that never wait for each other, unless --no-overlap).  No data-path collective; one all-reduce of five scalars (RCCL for
N > 1) closes the timed region.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=131072)
    ap.add_argument("--horizon", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the small configs reported beside the headline")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one stream; the timed region itself carries the per-pass HIP events")
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
    return {
        "backward": "k_backward4<true, false, false, false, float>" if (not full and ntiles <= min(t.bw4_tiles, t.split_bw_tiles)) else
                    ("k_backward2<%s, float>" if ntiles <= t.split_bw_tiles else "k_backward<%s, float>") % fl,
        # <diagonal, shared reference, 2 speculated trials, states re-computed (the iterates of the run are rollouts), float32 states>
        "forward": "k_forward_split<true, false, float>" if small else
                   "k_forward<true, false, 2, %s, float>" % ("true" if t.fw_recompute else "false"),
        "linesearch_update": "k_ls_final_split<true, false, float>" if small else "k_ls_final<true, false, float>",
        "linesearch_search": ("phase: k_ls_init_wl, k_ls_plan_wl, k_ls_trial_wl<true, false, %d, pinned|plain> x2, k_ls_replan" % max(t.ls_cpl, 1)) if wl
        else "phase: k_ls_init, k_ls_plan, k_ls_trial*, k_ls_resolve (round-based search)",
    }


def oracle_sample(pr, x0, iters, cores):
    """The oracle on `cores` threads over the first trajectories of the shard: (n, seconds, final XI, UI, history)."""
    from oracle import oracle as orc
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = orc.params(stepsize_0=1.0, armijo_maxiters=10)
    mdl = orc.default_model(pr.dt)
    nb = min(4 * cores, x0.shape[0])       # calibrate on 4 trajectories per core
    XI, UI = orc.initial_guess_batch(mdl, pr.xx_ref, x0[:nb], nthreads=cores)
    t0 = time.time()
    orc.newton_iterate_batch(op, prm, XI, UI, XI[:, :, 0].copy(), 0, 1, nthreads=cores)
    rate = nb / max(time.time() - t0, 1e-4)          # trajectory-iterations per second
    budget = 15.0 if cores > 1 else 5.0              # seconds of wall time for the sample
    n = int(min(x0.shape[0], 65536, max(cores, rate * 2.7 * budget / iters)))   # later iterations search longer
    XI, UI = orc.initial_guess_batch(mdl, pr.xx_ref, x0[:n], nthreads=cores)
    t0 = time.time()
    h = orc.newton_iterate_batch(op, prm, XI, UI, XI[:, :, 0].copy(), 0, iters, nthreads=cores)
    return n, time.time() - t0, XI, UI, h


def secondary_configs(torch, batch, problems, T):
    """The small BASELINE configs at their per-GPU size, beside the headline (a few seconds): configs[1] = 4096
    perturbed step-maneuver trajectories, 10 Newton iterations from the P-controller guess; configs[4] = 1024
    receding-horizon instances x 50 warm-started re-solves of 2 Newton iterations feeding tracking gains."""
    import time
    from aircraftoptimalcontrol_amd import mpc
    out = {}
    pr = problems.step_maneuver(tf=1.0, dt=1.0 / T)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    Bs = 4096
    s = batch.NewtonBatchSolver(bp, Bs, batch.make_params(stepsize_0=1.0, armijo_maxiters=10))
    x0 = problems.perturbed_x0(pr, Bs, seed=20260401)
    best = None
    for rep in range(3):
        s.set_initial_from_x0(x0)
        s.ntrials.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for kk in range(10):
            s.iterate(kk)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    out["configs[1]"] = {"workload": "4096 step-maneuver trajectories (perturbed x0), T=%d, fp64, Newton iterations kk=0..9" % T,
                         "ms_per_iteration": best / 10 * 1e3, "trajectory_iterations_per_s": Bs * 10 / best}
    steps, Bm = 50, 1024
    L = T + steps + 10
    full = problems.step_maneuver(tf=1.0, dt=1.0 / L)
    prm_ = problems.ProblemData("mpc", full.QQt, full.RRt, full.QQT, full.xx_ref, full.uu_ref, full.tt, full.tf, full.dt)
    rh = mpc.RecedingHorizon(prm_, problems.tracking_weights(), Bm, T, n_newton=2,
                             sigma=np.array([0.02, 0.02, 0.02, 0.002, 0.004, 0.002]))
    rh.start(problems.perturbed_x0(prm_, Bm, seed=1), cold_iters=10)
    rh.step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        rh.step(fetch=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out["configs[4]"] = {"workload": "1024 receding-horizon instances (per-GPU share of 8192), T=%d, %d re-solves x 2 Newton "
                                     "iterations + tracking gains + plant step, all on the device" % (T, steps),
                         "ms_per_receding_horizon_step": dt / steps * 1e3, "instance_steps_per_s": Bm * steps / dt}
    return out


def cpu_baseline(pr, x0, iters):
    """cpu_baseline block + what rel_err_vs_oracle needs.  Threads: the CPU share of a one-GPU box is 16 cores."""
    vis = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(vis, int(os.environ.get("AOC_CPU_THREADS", "16"))))
    n, dt, XI, UI, h = oracle_sample(pr, x0, iters, cores)
    n1, dt1, _, _, _ = oracle_sample(pr, x0, iters, 1)
    blk = {"value": n * iters / dt, "unit": "trajectory-Newton-iterations/s", "cores": cores, "kind": "port",
           "sample": "first %d trajectories x %d iterations of the same workload, oracle/aoc_oracle.c (C port of the "
                     "reference algorithm, fp64) with OpenMP over trajectories, %.1f s" % (n, iters, dt),
           "value_1thread": n1 * iters / dt1,
           "sample_1thread": "first %d trajectories x %d iterations, 1 thread, %.1f s" % (n1, iters, dt1)}
    return blk, (n, XI, UI, h)


def rel_err_vs_oracle(batch, bp, prm, x0, sample, iters):
    """The "fp64 rel-err" half of the metric: the same trajectories, the same iterations kk = 0..iters-1 on the GPU
    (free-running, results do not depend on the batch a trajectory is solved in) against the oracle's."""
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
    fin = np.isfinite(J).all(1) & np.isfinite(h["cost"]).all(1) & np.isfinite(ug).all((1, 2)) & np.isfinite(UO).all((1, 2))
    same = fin & (st == h["stepsize"]).all(1) & (nt == h["ntrials"]).all(1)      # identical Armijo histories
    cost_rel = (np.abs(J - h["cost"]) / np.abs(h["cost"]))[same].max(1) if same.any() else np.zeros(0)
    d = np.abs(ug - UO)
    chan = (d.max(2) / np.maximum(np.abs(UO).max(2), 1e-3)).max(1)                 # per input channel: max_t |du| / max_t |u|
    elem = (d / np.maximum(np.abs(UO), 1e-3)).max((1, 2))                          # SURVEY 8c gate: elementwise, floor 1e-3
    xsame = np.array([np.array_equal(xg[b], XO[b]) for b in range(n)])
    g = lambda a, f: float(f(a)) if a.size else None
    return {"n": int(n), "iterations": int(iters), "finite": int(fin.sum()),
            "identical_step_and_trial_history": int(same.sum()),
            "cost_rel_max": g(cost_rel, np.max), "cost_rel_median": g(cost_rel, np.median),
            "u_rel_channel_max": g(chan[same & xsame], np.max), "u_rel_channel_median": g(chan[same & xsame], np.median),
            "u_rel_elementwise_floor1e-3_max": g(elem[same & xsame], np.max),
            "states_bit_identical": int((same & xsame).sum()),
            "note": "u errors over trajectories whose Armijo history and float32 state trajectory equal the oracle's; the "
                    "others differ by one float32 rounding flip of a state (DESIGN.md §2)"}


def run(a):
    import torch
    from aircraftoptimalcontrol_amd import batch, problems, sharding

    rank, local_rank, world = sharding.env_rank_world()
    # Rehearsal knobs (not used by the driver): AOC_BENCH_BACKEND=gloo runs the collective over gloo on host copies,
    # AOC_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 (several ranks on a one-GPU box).
    backend = os.environ.get("AOC_BENCH_BACKEND", "nccl")
    one_dev = os.environ.get("AOC_BENCH_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if (world == 1 or one_dev) else local_rank)
    torch.cuda.set_device(dev)
    sharding.init_process_group(backend, dev)

    Bg, T, K = a.batch_per_gpu, a.horizon, a.steps
    pr = problems.step_maneuver(tf=1.0, dt=1.0 / T)
    assert pr.T == T
    first, n_own = sharding.shard_range(rank, world, Bg * world)       # weak scaling: Bg per rank
    assert n_own == Bg
    x0 = problems.random_x0(Bg, seed=20260403, first=first)            # synthetic inputs of this rank's shard
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt, device=dev)
    prm = batch.make_params(max_iters=200, stepsize_0=1.0, cc=0.5, beta=0.7, armijo_maxiters=10)
    s = batch.NewtonBatchSolver(bp, Bg, prm)       # one stream: the attribution pass, or everything with --no-overlap
    # two half batches on two streams pay while each half is still a large-batch launch (one wavefront per tile kernels)
    overlap = not a.no_overlap and s.nt >= 2048
    s2 = batch.TwoStreamNewtonSolver(bp, Bg, prm) if overlap else None
    x0d = torch.from_numpy(x0).to(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    def summary(sv):
        """scalar summary of the shard + the path's only collective (RCCL for N > 1)"""
        return sharding.reduce_summary(sharding.local_summary(*sv.summary_tensors()))

    def timed_region(sv, step):
        sv.set_initial_from_x0(x0d)
        barrier()
        t0 = time.perf_counter()
        evs = [step(k) for k in range(K)]
        summ = summary(sv)
        barrier()
        el = time.perf_counter() - t0
        return float(sharding.all_reduce(torch.tensor([el], dtype=torch.float64, device=dev), "max").item()), summ, evs

    # warmup: W iterations from the initial guess (and one summary, so that no lazily loaded code object is first
    # touched inside the timed region), then reset: the timed region is exactly iterations 0..K-1 of the solve
    for sv in ([s2, s] if overlap else [s]):
        sv.set_initial_from_x0(x0d)
        for k in range(a.warmup):
            sv.iterate(k)
        summary(sv)
    s.iterate_timed(0)
    if overlap:
        el, summ, _ = timed_region(s2, lambda k: s2.iterate(k))
        _, _, evs = timed_region(s, lambda k: s.iterate_timed(k))      # attribution: same iterations, one stream
    else:
        el, summ, evs = timed_region(s, lambda k: s.iterate_timed(k))
    res = s2 if overlap else s
    sc = res.scalars()
    if os.environ.get("AOC_BENCH_DUMP"):   # tests: this rank's per-trajectory results after the K iterations
        xx, uu = res.current()
        np.savez(os.path.join(os.environ["AOC_BENCH_DUMP"], "rank%d_of_%d.npz" % (rank, world)), first=first, xx=xx, uu=uu,
                 summary=summ.cpu().numpy(), **sc)

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
    dom = max((e for e in kernels if "algorithmic_bytes" in e), key=lambda e: e["avg_ms"])
    fin_b = int(summ[3].item() - summ[4].item())
    out = {
        "metric": "Newton iters/sec (whole node), batched 6-state T=%d trajectories; fp64 rel-err" % T,
        "value": Bg * world * K / el,
        "unit": "trajectory-Newton-iterations/s",
        "n_gpus": world, "steps": K, "warmup": a.warmup,
        "ms_per_step": el / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: step-maneuver, T=%d (tf=1, dt=%g), fp64, random x0 by global index, "
                               "P-controller initial guess, %d trajectories per GPU (2^20 over 8), Newton iterations "
                               "kk=0..%d from the initial guess (%d Gauss-Newton, %d full-Hessian; fixed-iteration mode: "
                               "nobody is stopped, so from kk~13 a third of the trajectories exhausts every line search)"
                               % (T, pr.dt, Bg, K - 1, int((~full).sum()), int(full.sum())),
                   "batch_per_gpu": Bg, "global_batch": Bg * world, "T": T, "parallelism": "batch-sharded x%d" % world,
                   "streams": "two half batches on two HIP streams (batch.TwoStreamNewtonSolver)" if overlap else "one",
                   "armijo": {"stepsize_0": 1.0, "cc": 0.5, "beta": 0.7, "maxiters": 10}},
        "roofline": {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["algorithmic_GBps"], "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": dom["frac_algorithmic"], "traffic": dom["traffic"],
                     "algorithmic_bytes_per_launch": dom["algorithmic_bytes"], "stored_bytes_per_launch": dom["stored_bytes"],
                     "frac_stored": dom["frac_stored"], "avg_launch_ms": dom["avg_ms"], "launches": dom["launches"],
                     "measured_in": ("one-stream repeat of the same %d iterations right after the timed region" % K)
                     if overlap else "the timed region"},
        "kernels": kernels,
        "attribution_ms_per_step": float(ms.sum(1).mean()),
        # whole iteration, per GPU: SURVEY 8d's 496 B per trajectory-stage over the wall time of a step
        "iteration_hbm_frac_per_gpu": ITERATION_BYTES * units * K / el / 1e9 / HBM_PEAK_GBS,
        "last_iter_mean_armijo_trials": float(summ[2].item() / summ[3].item()),
        "final_mean_cost_finite": float(summ[0].item() / max(fin_b, 1)),
        # trajectories whose cost is NaN/Inf: at kk = 9 the reference switches to the full Hessian (optcon.py:443) and
        # diverges on the same trajectories (checked against the oracle); throughput without them:
        "n_nonfinite": int(summ[4].item()),
        "value_finite_only": fin_b * K / el,
        "status_or_rank0": int(np.bitwise_or.reduce(sc["status"])),
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            it = min(K, 10)
            out["cpu_baseline"], sample = cpu_baseline(pr, x0, it)
            out["rel_err_vs_oracle"] = rel_err_vs_oracle(batch, bp, prm, x0, sample, it)
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out.setdefault("cpu_baseline", None)
            out["cpu_baseline_error"] = repr(e)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0 and world == 1 and not a.no_secondary and a.batch_per_gpu == 131072:
        try:
            out["secondary"] = secondary_configs(torch, batch, problems, a.horizon)
        except Exception as e:  # a report beside the headline, never a reason to lose the bench line
            out["secondary_error"] = repr(e)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


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
