"""How a logical env batch is split over the GPUs of a node.

Environments are independent (the reference already runs them as separate
processes against separate simulator instances,
``/root/reference/gym_roboy/train_parallel.py:19-29``), so there is no data-path
exchange: rank ``r`` of ``world`` owns the contiguous block
``[r*n/world, (r+1)*n/world)`` and keys every random stream by the *global* env
id, which makes results independent of ``world``.  The only collective is the
sum of the 8-double episode-statistics vector (``rb_env_stats``) over ranks:
RCCL over xGMI on the GPUs, gloo in the CPU tests.
"""

STAT_KEYS = ("sum_return", "sum_return_sq", "n_episodes", "sum_length", "n_goal_reached",
             "n_infeasible_steps", "n_env_steps", "sum_reward")


def shard_bounds(n_total: int, world: int, rank: int):
    """Contiguous block of rank ``rank``: (first global env id, count)."""
    if not (0 <= rank < world) or n_total < 0:
        raise ValueError("need 0 <= rank < world and n_total >= 0")
    base, extra = divmod(n_total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def allreduce_stats(stats, dist=None):
    """Sum a rank's statistics tensor over all ranks (in place); no-op without
    an initialised process group."""
    if dist is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    return stats


def summarize(stats) -> dict:
    """Mean return / length / rates from the summed vector."""
    s = [float(x) for x in stats]
    d = dict(zip(STAT_KEYS, s))
    ep = max(d["n_episodes"], 1.0)
    d["mean_return"] = d["sum_return"] / ep
    d["std_return"] = max(d["sum_return_sq"] / ep - d["mean_return"] ** 2, 0.0) ** 0.5
    d["mean_length"] = d["sum_length"] / ep
    d["goal_rate"] = d["n_goal_reached"] / ep
    d["infeasible_rate"] = d["n_infeasible_steps"] / max(d["n_env_steps"], 1.0)
    return d
