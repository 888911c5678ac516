"""Host-side logic that needs no GPU: problem builders against the reference's own arrays, the
synthetic-input generators, and the multi-process sharding/reduction over gloo (world_size 2)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from aircraftoptimalcontrol_amd import problems, sharding


@pytest.mark.parametrize("name,build", [("problem_step_T1000", lambda: problems.step_maneuver()),
                                        ("problem_step_T500", lambda: problems.step_maneuver(1.0, 2e-3)),
                                        ("problem_acro_T1000", lambda: problems.acrobatic())])
def test_problem_builders_equal_reference_arrays(name, build):
    g = load_golden(name)
    pr = build()
    for k in ("QQt", "RRt", "QQT", "xx_ref", "uu_ref"):
        assert np.array_equal(getattr(pr, k), g[k]), k
    assert np.array_equal(problems.XXE, g["xxe"]) and np.array_equal(problems.UUE, g["uue"][:2])
    assert pr.T == int(g["TT"])


def test_tracking_weights():
    Q, R, QT = problems.tracking_weights()
    g = load_golden("g4_lqr_tracking")
    assert np.array_equal(Q, g["QQt"]) and np.array_equal(R, g["RRt"]) and np.array_equal(QT, g["QQT"])


def test_random_x0_is_keyed_by_global_index():
    full = problems.random_x0(10000, seed=20260403, first=0)
    for first, n in ((0, 100), (4095, 3), (4096, 5000), (7777, 2223)):
        assert np.array_equal(problems.random_x0(n, seed=20260403, first=first), full[first:first + n])
    lo = np.array([-1, -1, 12, -0.2, -0.5, -0.2]); hi = np.array([1, 1, 20, 0.2, 0.5, 0.2])
    assert np.all(full >= lo) and np.all(full <= hi)


def test_oracle_initial_guess_batch_matches_reference_to_fp32_noise():
    from oracle import oracle as orc
    pr = problems.step_maneuver(1.0, 2e-3)
    g = load_golden("g9_minibatch_step_T500")
    XI, UI = orc.initial_guess_batch(orc.default_model(pr.dt), pr.xx_ref, g["x0"], nthreads=2)
    assert np.abs(XI - g["xx_init"]).max() < 1e-4 and np.abs(UI - g["uu_init"]).max() < 1e-3
    assert np.array_equal(XI[:, :, 0], g["x0"])


def test_shard_range_partitions():
    for world in (1, 2, 3, 8):
        for B in (1, 7, 64, 1000, 2 ** 20):
            parts = [sharding.shard_range(r, world, B) for r in range(world)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == B
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1


WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from aircraftoptimalcontrol_amd import problems, sharding
rank, local, world = sharding.env_rank_world()
assert sharding.init_process_group("gloo") == world
B = 1001
first, n = sharding.shard_range(rank, world, B)
x0 = problems.random_x0(n, seed=20260403, first=first)
# stand-in per-trajectory results: any function of x0 keyed by the global index
cost = x0[:, 2] ** 2 + x0[:, 0]
if rank == 1: cost[3] = np.nan
descent = -np.abs(x0[:, 4])
ntr = (np.arange(first, first + n) %% 5 + 1)
red = sharding.reduce_summary(sharding.local_summary(cost, descent, ntr))
# the same through torch tensors (what bench.py hands over) and the max-reduction of the timing
redt = sharding.reduce_summary(sharding.local_summary(torch.from_numpy(cost), torch.from_numpy(descent), torch.from_numpy(ntr)))
tmax = sharding.all_reduce(np.array([1.0 + rank]), "max")
np.save(os.path.join(os.environ["OUT"], "r%%d.npy" %% rank), np.concatenate([red, redt.numpy(), tmax, [first, n]]))
torch.distributed.destroy_process_group()
'''


def test_two_rank_gloo_reduction_equals_single_process(tmp_path):
    """8(e): shard the batch over 2 ranks, reduce the scalar summary with one all-reduce, and compare
    with the single-process sum over the whole batch.  (The same functions carry bench.py; two ranks through the
    real solver: tests/test_gpu_multirank.py.)"""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, OUT=str(tmp_path), MASTER_ADDR="127.0.0.1")
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)], env=env,
                          timeout=300)
    r0, r1 = np.load(tmp_path / "r0.npy"), np.load(tmp_path / "r1.npy")
    assert np.array_equal(r0[:11], r1[:11])               # both ranks hold the reduced vectors
    assert np.array_equal(r0[:5], r0[5:10]) and r0[10] == 2.0
    assert (r0[11], r0[12], r1[11], r1[12]) == (0, 501, 501, 500)
    B = 1001
    x0 = problems.random_x0(B, seed=20260403, first=0)
    cost = x0[:, 2] ** 2 + x0[:, 0]
    cost[501 + 3] = np.nan
    ref = sharding.local_summary(cost, -np.abs(x0[:, 4]), np.arange(B) % 5 + 1)
    assert np.allclose(r0[:5], ref, rtol=1e-13, atol=0)
    assert r0[3] == B and r0[4] == 1


def test_vector_issue_of_the_bench_line():
    """roofline.vector_issue: the dominant kernel priced against the fp64 issue roof from the committed SQ counters
    (profiles/r05_sq_counters.txt: 624.2 vector instructions per wavefront and stage of the large-batch forward pass)."""
    sys.path.insert(0, ROOT)
    import bench
    units = 131072 * 500
    v = bench.vector_issue({"kernel": "k_forward<true, false, 2, true, float>", "avg_ms": 1.8}, units, 500)
    assert v["valu_insts_per_launch"] == pytest.approx(6.379e8, rel=2e-4)          # the counter's own per-launch figure
    assert v["issue_ms_at_2.4GHz"] == pytest.approx(1.038, abs=1e-3) and v["frac"] == pytest.approx(0.577, abs=1e-3)
    assert bench.vector_issue({"kernel": "k_forward_split<true, false, float, 1>", "avg_ms": 0.3}, units, 500) is None


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """bench.py --gpus N must run N ranks: under a launcher with another WORLD_SIZE it exits non-zero instead of
    printing a line for the wrong rank count (no GPU is touched before that check)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()


def test_philox_known_answers_and_noise_draws_are_shard_invariant():
    """The checker of the device's disturbance model (mpc.noise_draws): its Philox4x32-10 against the known-answer vectors
    published with the generator (Random123 kat_vectors: zero counter and key; all ones; the digits of pi), and the draws
    as a function of the global instance index."""
    from aircraftoptimalcontrol_amd.mpc import philox4x32_10, noise_draws
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = philox4x32_10(*[[c] for c in ctr], *[[k] for k in key])
        assert tuple(int(g[0]) for g in got) == want
    sig = np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])
    a = noise_draws(12345678901234, 17, 0, 300, sig)
    assert np.array_equal(a[100:], noise_draws(12345678901234, 17, 100, 200, sig))
    assert not np.array_equal(a, noise_draws(12345678901234, 18, 0, 300, sig)) and np.isfinite(a).all()
    z = noise_draws(3, 0, 0, 100000, np.ones(6))
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01


# ---- the N > 1 control flow of bench.run with the solver stubbed out ------------------------------
STUB_WORKER = r'''
import json, os, sys, time, types
sys.path.insert(0, %r)
import numpy as np, torch
import torch.distributed as dist
from aircraftoptimalcontrol_amd import sharding, problems

rank = int(os.environ["RANK"])
LOG = []
# every collective this rank issues, in order: (what, payload shape) — a rank that issues another sequence than its peers
# is a deadlock (or a wrong reduction) on the driver's 8-GPU node
_ar, _bar = sharding.all_reduce, dist.barrier
def all_reduce(vec, op="sum"):
    LOG.append(["all_reduce", op, list(np.shape(vec))])
    return _ar(vec, op)
def barrier(*a, **k):
    LOG.append(["barrier"])
    return _bar(*a, **k)
sharding.all_reduce = all_reduce
sharding.reduce_summary = lambda vec: all_reduce(vec, "sum")
dist.barrier = barrier

# torch.cuda without a GPU: what run() asks of it.  Free memory differs by rank (each rank of a real node sees its own
# device; ranks rehearsing on one device see each other's transient allocations)
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda *a, **k: None
torch.cuda.mem_get_info = lambda *a, **k: (int((40 + 60 * rank) * 2**30), 288 * 2**30)

class Ev:
    def elapsed_time(self, other): return 1.0

class Stub:
    """the interface bench.run uses of NewtonBatchSolver / TwoStreamNewtonSolver; `iterate` takes a rank-dependent time"""
    PASSES = ("backward", "forward", "linesearch_search", "linesearch_update")
    made = 0
    def __init__(self, bp, B, prm, two=False):
        Stub.made += 1
        self.B, self.nt, self.two, self.k = B, (B + 63) // 64, two, 0
        self.problem = types.SimpleNamespace(device=torch.device("cpu"))
        self.cost = 0.001 * (1 + (Stub.made * 7 + rank * 3) %% 5)      # "placement": differs per candidate and per rank
    def set_initial_from_x0(self, x0): self.k = 0
    def iterate(self, kk=None): time.sleep(self.cost * 0.2); self.k += 1
    def iterate_timed(self, kk=None): self.iterate(kk); return [Ev() for _ in range(5)]
    def join(self): pass
    def summary(self, out=None): return torch.tensor([1.0 * self.B, -2.0 * self.B, 3.0 * self.B, float(self.B), 0.0], dtype=torch.float64)
    def scalars(self): return {"status": np.zeros(self.B, np.int32), "cost": np.zeros(self.B)}
    def current(self): return np.zeros((self.B, 6, 3)), np.zeros((self.B, 2, 3))

import aircraftoptimalcontrol_amd.batch as real_batch
stub = types.ModuleType("aircraftoptimalcontrol_amd.batch")
stub.NewtonBatchSolver = Stub
stub.TwoStreamNewtonSolver = lambda bp, B, prm: Stub(bp, B, prm, two=True)
stub.BatchProblem = lambda *a, **k: types.SimpleNamespace(device=torch.device("cpu"))
stub.make_params = real_batch.make_params
stub.ntiles = real_batch.ntiles
stub._dev_f64 = lambda a, d: a
stub._torch = lambda: torch
def best_placed(make_solver, x0, candidates=5, probe_iters=6, keep_first=False, force=None):
    real_batch._torch = lambda: torch          # the real function, on the stub solvers (its only torch.cuda call is patched above)
    real_batch._dev_f64 = lambda a, d: a
    return real_batch.best_placed(make_solver, x0, candidates, 2, keep_first, force)
stub.best_placed = best_placed
sys.modules["aircraftoptimalcontrol_amd.batch"] = stub
import aircraftoptimalcontrol_amd
aircraftoptimalcontrol_amd.batch = stub

import bench
bench.kernel_names = lambda nt, full: {p: p for p in Stub.PASSES}
bench.provenance = lambda: {}
a = bench.parse.__globals__["argparse"].Namespace(gpus=int(os.environ["WORLD_SIZE"]), steps=3, warmup=1, batch_per_gpu=int(os.environ["STUB_B"]),
        global_batch=0, cpu_budget_s=1.0, horizon=500, no_cpu_baseline=True, no_secondary=True, no_overlap=os.environ.get("STUB_NO_OVERLAP") == "1",
        placement_candidates=3)
try:
    bench.run(a)
finally:
    json.dump(LOG, open(os.path.join(os.environ["OUT"], "collectives_rank%%d.json" %% rank), "w"))
'''


@pytest.mark.parametrize("B,no_overlap", [(131072, "0"), (131072, "1"), (4096, "0")])
def test_every_rank_issues_the_same_sequence_of_collectives(tmp_path, B, no_overlap):
    """bench.run's control flow on four gloo ranks with the solver stubbed out: every rank must issue the same sequence of
    collectives whatever its own placement draw (AOC_BENCH_FORCE_CHOICE makes the ranks choose differently), its own free
    memory (differs by rank here) and its own timings.  Round 4's two deadlocks were rank-dependent branches around
    collectives (EXPERIMENTS.md); this is the cheap insurance for the driver's SCALE leg."""
    script = tmp_path / "stub_worker.py"
    script.write_text(STUB_WORKER % ROOT)
    port = 29600 + (B // 4096 + int(no_overlap)) % 50
    env = dict(os.environ, OUT=str(tmp_path), MASTER_ADDR="127.0.0.1", AOC_BENCH_BACKEND="gloo", AOC_BENCH_DEVICE="cpu",
               AOC_BENCH_FORCE_CHOICE="1", STUB_B=str(B), STUB_NO_OVERLAP=no_overlap)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env, timeout=300,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    logs = [json.load(open(tmp_path / ("collectives_rank%d.json" % k))) for k in range(4)]
    assert len(logs[0]) >= 8                                     # barriers, timing all-reduces, the summary
    for k in range(1, 4):
        assert logs[k] == logs[0], (k, len(logs[k]), len(logs[0]))
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 4 and line["collective"]["world_seen"] == 4
    assert line["value"] == pytest.approx(B * 4 * 3 / (line["ms_per_step"] * 3e-3))
