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


WORK_FIELDS = ("seed", "cell_id", "first_subframe", "nof_subframes")


def scatter_work(items=None, device=None, src=0):
    """The one exchange step of the multi-process path (SURVEY.md 8e / BASELINE north_star: "RCCL over xGMI only for the work-queue scatter"):
    rank `src` owns the work queue - one descriptor per rank, WORK_FIELDS as int64 (which synthetic cell / capture the rank replays and which
    subframe range of it) - and scatters it; every rank returns ITS descriptor as a dict.  dist.scatter on `device` tensors = ncclScatter-style
    send/recv over xGMI under the nccl (= RCCL) backend, TCP under gloo (tests/test_dist_gloo.py).  Not initialised: items[0].
    One capture spread over the GPUs of a node does NOT go through rank processes at all: lsn_phy_create_multi runs the engines of all devices in
    one process around one sequential search and hands chunks over through pinned host memory - the per-chunk descriptor
    exchange between rank processes that rounds 2-4 carried as an unused interface was removed in round 5."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return dict(zip(WORK_FIELDS, (int(v) for v in items[0])))
    rank, world = dist.get_rank(), dist.get_world_size()
    out = torch.zeros(len(WORK_FIELDS), dtype=torch.int64, device=device)
    if rank == src:
        assert items is not None and len(items) == world, "one work descriptor per rank"
        parts = [torch.tensor([int(v) for v in it], dtype=torch.int64, device=device) for it in items]
        dist.scatter(out, parts, src=src)
    else:
        dist.scatter(out, None, src=src)
    return dict(zip(WORK_FIELDS, (int(v) for v in out.cpu().tolist())))


def gather_flags(flag, device=None):
    """every rank contributes one small integer (e.g. "my cell's head equals the live oracle's": 1 / 0 / -1 = not checked) -> list over ranks, on every rank"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [int(flag)]
    t = torch.tensor([int(flag)], dtype=torch.int64, device=device)
    outs = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return [int(o.item()) for o in outs]
