"""Batch sharding over the GPUs of a node.

Trajectories are independent units: rank g of `world` owns the contiguous global index range
[g*Bg, (g+1)*Bg) (weak scaling: Bg per GPU fixed) or an even split of a fixed global batch (strong).
Nothing crosses GPUs inside an iteration; the only collective is one all-reduce(sum) of a handful of
fp64 scalars per reporting point (RCCL on GPUs, gloo in the CPU tests)."""
import numpy as np


def shard_range(rank, world, global_batch):
    """Contiguous even split of `global_batch` trajectories; the first (global_batch % world) ranks get one
    extra.  Returns (first, count)."""
    base, extra = divmod(int(global_batch), int(world))
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


SUMMARY_FIELDS = ("sum_cost", "sum_descent", "sum_trials", "n_traj", "n_nan")


def local_summary(cost, descent, ntrials):
    """fp64 vector of SUMMARY_FIELDS for this rank's shard (NaN costs are counted, not summed)."""
    cost = np.asarray(cost, dtype=np.float64)
    ok = np.isfinite(cost)
    return np.array([cost[ok].sum(), np.asarray(descent, dtype=np.float64)[ok].sum(),
                     float(np.asarray(ntrials).sum()), float(cost.size), float((~ok).sum())])


def reduce_summary(vec, device=None):
    """all-reduce(sum) over the default process group if one is initialised; identity otherwise."""
    import torch
    import torch.distributed as dist
    t = torch.as_tensor(np.asarray(vec, dtype=np.float64))
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()
