"""Batch sharding over the GPUs of a node — the one place where ranks meet.

Trajectories are independent units: rank g of `world` owns a contiguous range of global trajectory indices
(`shard_range`; weak scaling = the same count on every rank).  Nothing crosses GPUs inside an iteration; the
path's only collective is one all-reduce(sum) of SUMMARY_FIELDS (five fp64 scalars, 40 B) per reporting point:
RCCL over xGMI for GPU tensors (torch.distributed backend "nccl"), gloo for host tensors (CPU tests, one-GPU
rehearsals).  bench.py, tests/test_host_logic.py and tests/test_gpu_multirank.py all go through these functions.
"""
import os

import numpy as np

SUMMARY_FIELDS = ("sum_cost", "sum_descent", "sum_trials", "n_traj", "n_nonfinite")


def shard_range(rank, world, global_batch):
    """Contiguous even split of `global_batch` trajectories; the first (global_batch % world) ranks get one
    extra.  Returns (first, count)."""
    base, extra = divmod(int(global_batch), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def env_rank_world():
    """(rank, local_rank, world) as torchrun exports them; (0, 0, 1) for a plain `python` start."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend="nccl", device=None, force=False):
    """torch.distributed over RCCL ("nccl") or gloo when WORLD_SIZE > 1; no-op for one rank unless `force` (a one-rank
    group: what the one-GPU test of the RCCL branch uses).  Returns world.  An initialisation failure raises: there is
    no fallback to another backend."""
    import torch.distributed as dist
    _, _, world = env_rank_world()
    if (world > 1 or force) and not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    return world


def local_summary(cost, descent, ntrials):
    """SUMMARY_FIELDS of this rank's shard as an fp64 vector of the inputs' kind (torch tensors stay on their
    device, anything else goes through NumPy).  Non-finite costs are counted, not summed."""
    try:
        import torch
    except ImportError:  # pragma: no cover
        torch = None
    if torch is not None and isinstance(cost, torch.Tensor):
        ok = torch.isfinite(cost)
        z = torch.zeros_like(cost)
        n = torch.tensor(float(cost.numel()), dtype=torch.float64, device=cost.device)
        return torch.stack([torch.where(ok, cost, z).sum(), torch.where(ok, descent, z).sum(),
                            ntrials.sum().to(torch.float64), n, (~ok).sum().to(torch.float64)])
    cost = np.asarray(cost, dtype=np.float64)
    ok = np.isfinite(cost)
    return np.array([cost[ok].sum(), np.asarray(descent, dtype=np.float64)[ok].sum(),
                     float(np.asarray(ntrials).sum()), float(cost.size), float((~ok).sum())])


def all_reduce(vec, op="sum"):
    """all-reduce over the default process group (identity for one rank).  A GPU tensor is reduced in place by the
    group's backend when that is RCCL; with a gloo group (CPU tests, several ranks rehearsing on one GPU) it
    travels through a host copy.  Returns the same kind as given."""
    import torch
    import torch.distributed as dist
    is_t = isinstance(vec, torch.Tensor)
    t = vec if is_t else torch.as_tensor(np.asarray(vec, dtype=np.float64))
    if dist.is_available() and dist.is_initialized():   # a one-rank group reduces too (identity, but through the backend)
        rop = dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MAX
        if t.is_cuda and dist.get_backend() != "nccl":
            h = t.cpu()
            dist.all_reduce(h, op=rop)
            t = h.to(t.device)
        else:
            dist.all_reduce(t, op=rop)
    return t if is_t else t.cpu().numpy()


def reduce_summary(vec):
    """The path's one collective: all-reduce(sum) of a local_summary() vector."""
    return all_reduce(vec, "sum")
