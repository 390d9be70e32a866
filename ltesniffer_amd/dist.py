"""Multi-GPU helpers of the offline / file-replay path: one process per GPU (torch.distributed; backend nccl = RCCL on
ROCm, gloo in the CPU tests).  The path shards with NO data-path exchange: cells (or contiguous capture ranges) are
independent units, so the only collectives are the barrier and the MAX / SUM reductions of the timing."""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def rank_workload(config, rank):
    """seed and physical cell id of the synthetic cell replayed by `rank` (config 5 style: one capture per GPU)"""
    return dict(seed=3 + 50 * rank, cell_id=1 + rank)


def shard_ranges(n_subframes, world):
    """contiguous subframe ranges [lo, hi) per rank, multiples of 10 subframes (one frame) except possibly the last"""
    frames = (n_subframes + 9) // 10
    out, lo = [], 0
    for r in range(world):
        nf = frames // world + (1 if r < frames % world else 0)
        hi = min(n_subframes, lo + 10 * nf)
        out.append((lo, hi))
        lo = hi
    return out


def reduce_max_sum(value, count, device=None):
    """-> (max over ranks of value, sum over ranks of count); identity when not initialised"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value), float(count)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    c = torch.tensor([float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c.item())
