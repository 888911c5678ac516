"""SURVEY 8(e) on hardware that has one GPU: `bench.py --gpus 2` starts two ranks itself (both mapped to cuda:0,
collective over gloo), each runs the REAL solver on its shard; every trajectory must equal the single-process run
bit for bit, the all-reduced sums the single-process sums, and the line must say n_gpus: 2."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(tmp, gpus, per_gpu, steps=3, extra=("--cpu-budget-s", "1.5"), light=False, force_choice=False):
    d = tmp / ("n%d_%d" % (gpus, per_gpu))
    d.mkdir()
    env = dict(os.environ, AOC_BENCH_ONE_DEVICE="1", AOC_BENCH_BACKEND="gloo", AOC_BENCH_DUMP=str(d),
               AOC_BENCH_DUMP_LIGHT="1" if light else "0", AOC_BENCH_FORCE_CHOICE="1" if force_choice else "0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    try:    # what earlier tests of this process left in torch's cache is not available to the ranks otherwise
        import torch
        torch.cuda.empty_cache()
    except Exception:
        pass
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", str(steps),
                        "--warmup", "1", "--batch-per-gpu", str(per_gpu)] + list(extra), env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0]), [dict(np.load(d / ("rank%d_of_%d.npz" % (k, gpus)))) for k in range(gpus)]


def test_two_ranks_through_the_real_solver_equal_one_process(tmp_path):
    per = 8256                                  # 129 tiles per rank (above the horizon cut of small batches); the single process solves all 16 512 at once
    two, parts = _bench(tmp_path, 2, per)
    one, (whole,) = _bench(tmp_path, 1, 2 * per)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["global_batch"] == 2 * per == one["config"]["global_batch"]
    assert [int(p["first"]) for p in parts] == [0, per]
    for key in ("xx", "uu", "cost", "cost_new", "descent", "stepsize", "ntrials", "status"):
        both = np.concatenate([p[key] for p in parts])
        assert np.array_equal(both, whole[key], equal_nan=True), key
    # the path's one collective: both ranks hold the same reduced vector, equal to the single-process sums
    assert np.array_equal(parts[0]["summary"], parts[1]["summary"])
    assert np.allclose(parts[0]["summary"], whole["summary"], rtol=1e-12, atol=0)
    assert parts[0]["summary"][3] == 2 * per
    assert two["steps"] == 3 and two["value"] > 0 and two["scaling"] == "weak"
    # the N > 1 line is a full record and says what ran: the CPU baseline and the comparison with the
    # oracle come from rank 0 after the process group is gone, the collective is named and timed by itself
    cb = two["cpu_baseline"]
    assert cb is not None and cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "cpu_baseline_error" not in two
    re_ = two["rel_err_vs_oracle"]
    assert re_["n"] >= 1 and re_["iterations"] == 3 and re_["identical_step_and_trial_history"] >= 0.99 * re_["n"]
    assert re_["all_finite"]["u_rel_channel"]["median"] < 1e-9
    co = two["collective"]
    assert co["world_seen"] == 2 and co["backend"] == "gloo" and co["payload_bytes"] == 40 and co["us"] > 0
    lo, hi = two["per_rank_ms_per_step"]
    assert 0 < lo <= hi and abs(hi - two["ms_per_step"]) < 1e-9
    assert one["collective"] is None and len(one["per_rank_ms_per_step"]) == 2
    # the bench contract, key by key, on both lines
    for line in (one, two):
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert key in line, key
        assert line["higher_is_better"] is True and line["vs_baseline"] is None and line["dtype"] == "f64"
        assert line["data"] == "synthetic" and "workload" in line["config"] and "model" not in line["config"]
        assert line["metric"].startswith("Newton iters/sec") and line["unit"] == "trajectory-Newton-iterations/s"
        rf = line["roofline"]
        assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
        assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and "traffic" in rf and rf["kernel"].startswith("k_")
        assert set(("value", "unit", "cores", "kind", "sample")) <= set(line["cpu_baseline"])
        assert abs(line["value"] - line["config"]["global_batch"] * line["steps"] / (line["ms_per_step"] * line["steps"] * 1e-3)) < 1e-6 * line["value"]
    out = os.path.join(ROOT, "gpurun_out", "two_rank_rehearsal.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump({"two_ranks_one_device_gloo": two, "one_process": one}, open(out, "w"), indent=1)


def test_strong_scaling_option_splits_a_global_batch(tmp_path):
    """--global-batch G: the same total work over N ranks ("scaling": "strong"); each rank owns G / N contiguous global
    indices, so the shards of a 2-rank run are the halves of the 1-rank run."""
    G = 2 * 8256
    two, parts = _bench(tmp_path, 2, 0, steps=2, extra=("--global-batch", str(G), "--no-cpu-baseline"))
    assert two["scaling"] == "strong" and two["config"]["global_batch"] == G and two["config"]["batch_per_gpu"] == G // 2
    assert [int(p["first"]) for p in parts] == [0, G // 2] and parts[0]["xx"].shape[0] == G // 2
    assert two["cpu_baseline"] is None and two["collective"]["world_seen"] == 2
    # a global batch that does not divide is refused before anything runs
    env = dict(os.environ, AOC_BENCH_ONE_DEVICE="1", AOC_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--global-batch", "8321"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "not a multiple" in r.stderr


def test_two_rank_rehearsal_with_placement_candidates_at_the_per_gpu_size(tmp_path):
    """The code path eight ranks take on an 8-GPU node, executed once on the one-GPU box: two ranks of
    131 072 trajectories each (--global-batch 262144: the north-star shard size), every rank choosing its allocation among
    candidates within ITS share of the free memory (fits()), two half batches on two streams, the first-allocated solver
    timed beside the chosen one, every rank's draw gathered into the line, teardown, rank 0 finishing the record.  The
    ranks are MADE to choose differently (rank r takes candidate r): twice a branch that depended on a rank's own choice or
    on its own view of the free memory sat around a collective and hung two ranks that chose differently (round 4)."""
    G = 2 * 131072
    two, parts = _bench(tmp_path, 2, 0, steps=3, extra=("--global-batch", str(G), "--placement-candidates", "2",
                                                        "--no-cpu-baseline", "--no-secondary"), light=True, force_choice=True)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["batch_per_gpu"] == 131072
    assert "two half batches" in two["config"]["streams"]
    pt = two["placement_tuning"]
    assert pt["candidates"] == 2 and len(pt["two_stream_solver"]["ms_per_iteration"]) == 2
    assert pt["two_stream_solver"]["probe_wall_s"] > 0
    draws = np.array(pt["per_rank_probe_ms_chosen_first"])
    assert draws.shape == (2, 2) and (draws > 0).all()
    # rank 0 took its first candidate, rank 1 its second (the choice itself is checked, not two rounded timings that may coincide)
    assert pt["two_stream_solver"]["chosen"] == 0 and draws[0, 0] == draws[0, 1]
    assert pt["per_rank_chosen"] == [0, 1]
    assert two["ms_per_step_first_allocated"] > 0 and two["value_first_allocated"] > 0
    assert two["collective"]["world_seen"] == 2 and [int(p["first"]) for p in parts] == [0, 131072]
    assert parts[0]["cost"].shape[0] == 131072 and np.array_equal(parts[0]["summary"], parts[1]["summary"])
    out = os.path.join(ROOT, "gpurun_out", "two_rank_rehearsal_placement.json")
    json.dump(two, open(out, "w"), indent=1)


def test_streams_of_a_two_stream_solver_run_side_by_side():
    """The runtime maps HIP streams onto a few hardware queues and two streams on one queue take turns (of eight streams
    created in a row, five pairs shared a queue on this pool; a two-stream solver on such a pair ran at one-stream speed:
    the always-slow second candidate of round 3's placement probes).  aoc_streams_concurrent tells, batch.concurrent_streams
    makes sure: a stream is not concurrent with itself, the streams handed out are pairwise concurrent, and so are those
    of a TwoStreamNewtonSolver — also of the fourth one built in a process."""
    import ctypes as C
    import torch
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    from aircraftoptimalcontrol_amd._lib import lib
    conc = lambda a, b: lib().aoc_streams_concurrent(C.c_void_p(a.cuda_stream), C.c_void_p(b.cuda_stream))
    s = torch.cuda.Stream()
    assert conc(s, s) == 0
    three = aoc.concurrent_streams("cuda:0", 3)
    assert len(three) == 3 and all(conc(a, b) == 1 for i, a in enumerate(three) for b in three[i + 1:])
    cur = torch.cuda.current_stream()
    pair = aoc.concurrent_streams("cuda:0", 2, first=cur)
    assert pair[0] is cur and conc(pair[0], pair[1]) == 1
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    for _ in range(4):
        t = aoc.TwoStreamNewtonSolver(bp, 256, aoc.make_params(stepsize_0=1.0, armijo_maxiters=10))
        assert conc(*t.streams) == 1


def test_two_stream_solver_equals_one_stream():
    """batch.TwoStreamNewtonSolver (two half batches on two HIP streams that never wait for each other — what
    bench.py times) against one NewtonBatchSolver: iterates, steps, trial counts and costs bit for bit, across the
    Hessian switch, on a ragged batch."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 66000 + 37
    x0 = problems.random_x0(B, seed=20260403)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    one = aoc.NewtonBatchSolver(bp, B, prm)
    one.set_initial_from_x0(x0)
    one.run_fixed(11, record=False)
    two = aoc.TwoStreamNewtonSolver(bp, B, prm)
    assert two.Ba % 64 == 0 and 0 < two.Ba < B
    two.set_initial_from_x0(x0)
    two.run_fixed(11)
    a, b = one.scalars(), two.scalars()
    for key in ("stepsize", "ntrials", "cost", "cost_new", "descent", "status"):
        assert np.array_equal(a[key], b[key], equal_nan=True), key
    (xa, ua), (xb, ub) = one.current(), two.current()
    assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True)
    ja, da, na = (t.cpu().numpy() for t in one.summary_tensors())
    jb, db, nb = (t.cpu().numpy() for t in two.summary_tensors())
    assert np.array_equal(ja, jb, equal_nan=True) and np.array_equal(da, db, equal_nan=True) and np.array_equal(na, nb)
    # the same iterations issued pass by pass with the halves taking turns in their search rounds (iterate_phased:
    # scheduling only — measured slower than free-running halves, off by default)
    two.phased = True
    two.set_initial_from_x0(x0)
    two.run_fixed(11)
    c = two.scalars()
    for key in ("stepsize", "ntrials", "cost", "cost_new", "descent", "status"):
        assert np.array_equal(a[key], c[key], equal_nan=True), key
    xc, uc = two.current()
    assert np.array_equal(xa, xc, equal_nan=True) and np.array_equal(ua, uc, equal_nan=True)


def test_rccl_branch_on_one_gpu(tmp_path):
    """The branch the 8-GPU run takes — dist.init_process_group("nccl", device_id=...), the in-place RCCL all-reduce of a
    DEVICE tensor, dist.barrier() under nccl — cannot run with two ranks on one GPU (RCCL refuses a duplicate device), so
    it is exercised with a one-rank group in a child process: RCCL loads, the communicator comes up on cuda:0, the
    summary vector goes through ncclAllReduce and comes back unchanged, the group is torn down."""
    code = r"""
import os, sys, socket
sys.path.insert(0, %r)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from aircraftoptimalcontrol_amd import sharding
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
sharding.init_process_group("nccl", dev, force=True)
assert dist.is_initialized() and dist.get_backend() == "nccl" and dist.get_world_size() == 1
cost = torch.tensor([1.5, float("nan"), 2.5], dtype=torch.float64, device=dev)
vec = sharding.local_summary(cost, torch.tensor([-1.0, -2.0, -3.0], dtype=torch.float64, device=dev), torch.tensor([1, 2, 3], dtype=torch.int32, device=dev))
ref = vec.clone()
out = sharding.reduce_summary(vec)
torch.cuda.synchronize(dev)
assert out.is_cuda and out.data_ptr() == vec.data_ptr()          # reduced in place on the device, by RCCL
assert torch.equal(out, ref) and out.tolist() == [4.0, -4.0, 6.0, 3.0, 1.0]
mx = sharding.all_reduce(torch.tensor([3.25], dtype=torch.float64, device=dev), "max")
assert mx.item() == 3.25
dist.barrier(); torch.cuda.synchronize(dev)
dist.destroy_process_group()
print("RCCL_ONE_RANK_OK")
""" % ROOT
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_best_placed_picks_a_solver_and_changes_no_result():
    """batch.best_placed (allocation autotuning: build a few solvers, time a few iterations, keep the fastest) returns a
    working solver of the kind asked for, reports one timing per candidate, and — the buffers holding the same values
    wherever they were allocated — changes no result."""
    from aircraftoptimalcontrol_amd import batch as aoc, problems
    pr = problems.step_maneuver(1.0, 2e-3)
    bp = aoc.BatchProblem(pr.QQt, pr.RRt, pr.QQT, pr.xx_ref, pr.uu_ref, pr.dt)
    B = 300
    x0 = problems.random_x0(B, seed=5)
    prm = aoc.make_params(stepsize_0=1.0, armijo_maxiters=10)
    ref = aoc.NewtonBatchSolver(bp, B, prm)
    ref.set_initial_from_x0(x0)
    ref.run_fixed(4, record=False)
    for make in (lambda: aoc.NewtonBatchSolver(bp, B, prm), lambda: aoc.TwoStreamNewtonSolver(bp, B, prm)):
        sv, rep = aoc.best_placed(make, x0, candidates=3, probe_iters=2)
        assert len(rep["ms_per_iteration"]) == 3 and 0 <= rep["chosen"] < 3 and min(rep["ms_per_iteration"]) > 0
        assert rep["ms_per_iteration"][rep["chosen"]] == min(rep["ms_per_iteration"])
        sv.set_initial_from_x0(x0)
        sv.run_fixed(4) if isinstance(sv, aoc.TwoStreamNewtonSolver) else sv.run_fixed(4, record=False)
        (xa, ua), (xb, ub) = ref.current(), sv.current()
        assert np.array_equal(xa, xb, equal_nan=True) and np.array_equal(ua, ub, equal_nan=True)
        a, b = ref.scalars(), sv.scalars()
        for key in a:
            assert np.array_equal(a[key], b[key], equal_nan=True), key
