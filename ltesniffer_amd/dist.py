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


class DescriptorExchange:
    """The exchange step one capture spread over several RANK PROCESSES really has (SURVEY.md 8e "Collective"): the stateless DSP of a chunk of
    subframes runs on the rank that owns the chunk, the sequential FALCON search runs on ONE rank (`search_rank`).  Per chunk:

        owner --- candidate table + CCE powers + (cfi, snr) per subframe --->  search rank        (`tables_up`)
        owner <-- accepted-DCI descriptors of the chunk (6 words per DCI) ---  search rank        (`grants_down`)

    KB-scale messages, latency- not bandwidth-bound: one message per chunk and direction.  Built on torch.distributed point-to-point calls -
    ncclSend / ncclRecv over xGMI with the nccl (= RCCL) backend (tensors on `device`), TCP under gloo (the CPU test,
    tests/test_dist_exchange.py).  A chunk owned by the search rank itself never touches the network.  The single-process multi-GPU mode
    (lsn_phy_create_multi) makes the same hand-off through pinned host memory."""

    def __init__(self, search_rank=0, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.search_rank = search_rank
        self.device = device

    @staticmethod
    def owner_of(chunk, world):
        """chunks go round-robin to the ranks (contiguous chunks of >= 64 subframes amortise the launches)"""
        return chunk % world

    def _send(self, arr, dst):
        t = self.torch.from_numpy(arr.view("uint8").reshape(-1))
        if self.device is not None:
            t = t.to(self.device)
        n = self.torch.tensor([t.numel()], dtype=self.torch.int64, device=self.device)
        self.dist.send(n, dst)
        self.dist.send(t, dst)

    def _recv(self, src):
        n = self.torch.zeros(1, dtype=self.torch.int64, device=self.device)
        self.dist.recv(n, src)
        t = self.torch.empty(int(n.item()), dtype=self.torch.uint8, device=self.device)
        self.dist.recv(t, src)
        return t.cpu().numpy()

    def tables_up(self, chunk, payload=None):
        """owner: payload = bytes-like numpy array of the chunk's stage-A results -> returns it on the search rank, None elsewhere"""
        owner = self.owner_of(chunk, self.world)
        if owner == self.search_rank:
            return payload if self.rank == owner else None
        if self.rank == owner:
            self._send(payload, self.search_rank)
            return None
        if self.rank == self.search_rank:
            return self._recv(owner)
        return None

    def grants_down(self, chunk, payload=None):
        """search rank: payload = the chunk's accepted-DCI descriptors -> returns them on the chunk's owner, None elsewhere"""
        owner = self.owner_of(chunk, self.world)
        if owner == self.search_rank:
            return payload if self.rank == owner else None
        if self.rank == self.search_rank:
            self._send(payload, owner)
            return None
        if self.rank == owner:
            return self._recv(self.search_rank)
        return None
