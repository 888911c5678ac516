#!/usr/bin/env python3
"""bench.py — throughput of the batched Newton/LQR hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[3], the configuration the headline metric is quoted on): the
step-maneuver problem with tf = 1, dt = 2e-3 (T = 500), fp64, random x0 keyed by the global
trajectory index, P-controller initial guess, sharded 131 072 trajectories per GPU (2^20 over the 8
GPUs of a node; weak scaling: per-GPU work is fixed).  One "step" = one full Newton iteration
(backward Riccati pass, forward LQR pass with the first Armijo trial, Armijo back-tracking + update)
for every trajectory of the shard; the timed region runs iterations kk = 0..K-1 from the initial
guess with inputs resident in HBM.  No data-path collective; one RCCL all-reduce of four scalars
closes the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel, measured with HIP events on
the launch stream inside the timed region; `cpu_baseline` is the CPU oracle (a C port of the
reference's algorithm, oracle/aoc_oracle.c) timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# algorithmic bytes per trajectory and stage: fp64 elements x 8
# per kernel: fp64 elements the kernel's formulation moves; "iteration": SURVEY 8d's figure for a whole
# Newton iteration (its formulation also carries g = B^T lambda + r, 2 elements written and read)
ALGO_BYTES = {"backward": (8 + 14) * 8, "forward": (22 + 2) * 8, "linesearch": (4 + 8) * 8, "iteration": 496}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=131072)
    ap.add_argument("--horizon", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(pr, x0, iters):
    """The oracle on this host's cores: a bounded sample of the same workload, same iterations."""
    from oracle import oracle as orc
    op = orc.OracleProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    prm = orc.params(stepsize_0=1.0, armijo_maxiters=10)
    # threads: the CPU share of a one-GPU box is 16 cores, whatever the number of visible CPUs
    vis = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(vis, int(os.environ.get("AOC_CPU_THREADS", "16"))))
    # calibrate on 4 trajectories per core, then size the sample for ~15 s of wall time
    nb = min(4 * cores, x0.shape[0])
    mdl = orc.default_model(pr.dt)
    XI, UI = orc.initial_guess_batch(mdl, pr.xx_ref, x0[:nb], nthreads=cores)
    t0 = time.time()
    orc.newton_iterate_batch(op, prm, XI, UI, XI[:, :, 0].copy(), 0, 1, nthreads=cores)
    rate = nb / max(time.time() - t0, 1e-4)          # trajectory-iterations per second
    n = int(min(x0.shape[0], 65536, max(cores, rate * 40.0 / iters)))
    XI, UI = orc.initial_guess_batch(mdl, pr.xx_ref, x0[:n], nthreads=cores)
    t0 = time.time()
    orc.newton_iterate_batch(op, prm, XI, UI, XI[:, :, 0].copy(), 0, iters, nthreads=cores)
    dt = time.time() - t0
    return {"value": n * iters / dt, "unit": "trajectory-Newton-iterations/s", "cores": cores, "kind": "port",
            "sample": "first %d trajectories x %d iterations of the same workload, oracle/aoc_oracle.c (C port of "
                      "the reference algorithm, fp64) with OpenMP over trajectories, %.1f s" % (n, iters, dt)}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from aircraftoptimalcontrol_amd import batch, problems

    # Rehearsal knobs (not used by the driver): AOC_BENCH_BACKEND=gloo runs the collectives over gloo on
    # host copies, AOC_BENCH_ONE_DEVICE=1 maps every rank to cuda:0 (several ranks on a one-GPU box).
    backend = os.environ.get("AOC_BENCH_BACKEND", "nccl")
    one_dev = os.environ.get("AOC_BENCH_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if (world == 1 or one_dev) else local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    def all_reduce(t, op):
        if world == 1:
            return t
        if backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        h = t.cpu()
        dist.all_reduce(h, op=op)
        return h.to(t.device)

    Bg, T = a.batch_per_gpu, a.horizon
    pr = problems.step_maneuver(tf=1.0, dt=1.0 / T)
    assert pr.T == T
    # synthetic inputs of this rank's shard (global indices rank*Bg .. ), built on the host
    x0 = problems.random_x0(Bg, seed=20260403, first=rank * Bg)
    bp = batch.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt, device=dev)
    prm = batch.make_params(max_iters=200, stepsize_0=1.0, cc=0.5, beta=0.7, armijo_maxiters=10)
    s = batch.NewtonBatchSolver(bp, Bg, prm)
    x0d = torch.from_numpy(x0).to(dev)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def summary():
        """scalar summary of the shard + the path's only collective: one all-reduce(sum) (RCCL for N > 1)"""
        Jn = s.J[s.jcur][:Bg]
        ok = torch.isfinite(Jn)
        v = torch.stack([torch.where(ok, Jn, torch.zeros_like(Jn)).sum(),
                         torch.where(ok, s.descent[:Bg], torch.zeros_like(Jn)).sum(),
                         s.ntrials[:Bg].sum().to(torch.float64),
                         torch.tensor(float(Bg), dtype=torch.float64, device=dev),
                         (~ok).sum().to(torch.float64)])
        return all_reduce(v, dist.ReduceOp.SUM)

    # warmup: W iterations from the initial guess (and one summary, so that no lazily loaded code
    # object is first touched inside the timed region), then reset: the timed region is exactly
    # iterations 0..K-1 of the solve
    s.set_initial_from_x0(x0d)
    for k in range(a.warmup):
        s.iterate_timed(k)
    summary()
    s.set_initial_from_x0(x0d)
    barrier()
    t0 = time.perf_counter()
    evs = []
    for k in range(a.steps):
        evs.append(s.iterate_timed(k))
    summ = summary()
    barrier()
    el = time.perf_counter() - t0
    tmax = all_reduce(torch.tensor([el], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
    el = float(tmax.item())

    # per-kernel durations from the HIP events (this rank)
    names = ("backward", "forward", "linesearch")
    dur = {n: [] for n in names}
    for ev in evs:
        for i, n in enumerate(names):
            dur[n].append(ev[i].elapsed_time(ev[i + 1]))
    avg = {n: float(np.mean(v)) for n, v in dur.items()}
    dom = max(avg, key=lambda n: avg[n])
    units = Bg * T  # trajectory-stages one launch processes
    ach = ALGO_BYTES[dom] * units / (avg[dom] * 1e-3) / 1e9
    traffic = None
    tp = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tp):
        try:
            tj = json.load(open(tp))
            if tj.get("batch_per_gpu") == Bg and tj.get("T") == T:
                traffic = tj.get("kernels", {}).get(dom, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    sc = s.scalars()
    out = {
        "metric": "Newton iters/sec (whole node), batched 6-state T=%d trajectories" % T,
        "value": Bg * world * a.steps / el,
        "unit": "trajectory-Newton-iterations/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": el / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "BASELINE configs[3]: step-maneuver, T=%d (tf=1, dt=%g), fp64, random x0 by global "
                               "index, P-controller initial guess, %d trajectories per GPU (2^20 over 8), Newton "
                               "iterations kk=0..%d from the initial guess" % (T, pr.dt, Bg, a.steps - 1),
                   "batch_per_gpu": Bg, "global_batch": Bg * world, "T": T, "parallelism": "batch-sharded x%d" % world,
                   "armijo": {"stepsize_0": 1.0, "cc": 0.5, "beta": 0.7, "maxiters": 10}},
        "roofline": {"bound": "hbm", "kernel": "k_" + dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                     "algorithmic_bytes_per_launch": ALGO_BYTES[dom] * units, "avg_launch_ms": avg[dom]},
        "kernels_ms": avg,
        # whole iteration, per GPU: SURVEY 8d's 496 B per trajectory-stage over the wall time of a step
        "iteration_hbm_frac_per_gpu": ALGO_BYTES["iteration"] * units * a.steps / el / 1e9 / HBM_PEAK_GBS,
        "last_iter_mean_armijo_trials": float(summ[2].item() / summ[3].item()),
        "final_mean_cost_finite": float(summ[0].item() / max(summ[3].item() - summ[4].item(), 1.0)),
        # trajectories whose cost is NaN/Inf: at kk = 9 the reference switches to the full Hessian
        # (optcon.py:443) and diverges on the same trajectories (checked against the oracle)
        "n_nonfinite": int(summ[4].item()),
        "status_or_rank0": int(np.bitwise_or.reduce(sc["status"])),
    }
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(pr, x0, min(a.steps, 10))
        except Exception as e:  # the baseline is a report, never a reason to lose the bench line
            out["cpu_baseline"] = None
            out["cpu_baseline_error"] = repr(e)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
