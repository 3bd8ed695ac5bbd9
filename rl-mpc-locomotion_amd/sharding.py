"""Robot sharding across the GPUs of one node.

Robots are independent -- there is no cross-robot term anywhere on the path -- so the batch is cut into
contiguous blocks of robot indices, one per rank (one process per GPU), each rank keeping its robots'
persistent state.  No data-path collective is needed; the only optional exchange is an all-gather of the
per-robot torques ([n_local, 12] float32 per rank) when one consumer wants the whole batch on every
device.  ``torch.distributed`` with backend "nccl" is RCCL on ROCm; the CPU tests run the same code over
"gloo".
"""


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of rank `rank`; the first n_total % world ranks get one extra robot."""
    if not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(n_total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_total: int, world: int):
    return [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]


def all_gather_torques(local, n_total: int, group=None):
    """All-gather the per-rank torque blocks into a [n_total, 12] tensor on every rank (one collective;
    the message is 48 B per robot, i.e. latency-bound on xGMI -- SURVEY.md 5)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_total, world)
    if len(set(sizes)) == 1:
        out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # uneven shards: pad every block to the largest one (collectives want equal sizes), then drop the padding
    mx = max(sizes)
    padded = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded, group=group)
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def max_over_ranks(seconds: float, device, group=None) -> float:
    """bench.py's timing rule: the job is as slow as its slowest rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
