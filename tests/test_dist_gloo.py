"""CPU test of the N > 1 path: world_size-2 gloo group, cell-per-rank sharding with no data-path collective, MAX / SUM
reductions of the timing exactly as bench.py uses them; every rank's shard is decoded (by the oracle here - no GPU) and
the union equals the sequential run."""
import os
import sys

import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from ltesniffer_amd import dist as ld
    from lsn_testlib import OracleWorker, TxGen, parse_pcap, scenario
    dist.init_process_group("gloo", rank=rank, world_size=world)
    assert ld.env_rank_world() == (rank, world, rank)
    # the work queue lives on rank 0 and is scattered (bench.py --gpus N does exactly this over RCCL)
    work = ld.scatter_work([tuple(ld.rank_workload("small", r).values()) + (0, 12) for r in range(world)] if rank == 0 else None)
    assert work["first_subframe"] == 0 and work["nof_subframes"] == 12
    wl = dict(seed=work["seed"], cell_id=work["cell_id"])
    assert wl == ld.rank_workload("small", rank)
    sc = scenario("small", **wl)
    tx = TxGen(**sc)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"])
    n = 12
    for _ in range(n):
        tti, iq, _p = tx.next()
        ow.work(iq, tti)
    nrec = len(parse_pcap(ow.pcap_bytes()))
    dist.barrier()
    tmax, total = ld.reduce_max_sum(1.0 + rank, n)
    flags = ld.gather_flags(1 if nrec > 0 else 0)   # per-rank parity verdicts travel to every rank
    assert flags == [1] * world
    q.put((rank, wl["cell_id"], nrec, tmax, total, ld.shard_ranges(95, world)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_cell_sharding():
    world, port = 2, 29500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [1, 2]                  # one synthetic cell per rank
    assert all(r[2] > 0 for r in res)                     # every shard decoded PDUs
    assert all(r[3] == 2.0 and r[4] == 24.0 for r in res)  # MAX of the times, SUM of the subframes
    rng = res[0][5]
    assert rng == [(0, 50), (50, 95)] and rng == res[1][5]
