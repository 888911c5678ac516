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


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    """bench.py --gpus N must run N ranks: under a launcher with another WORLD_SIZE it exits non-zero instead of
    printing a line for the wrong rank count (no GPU is touched before that check)."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE" in r.stderr and not r.stdout.strip()
