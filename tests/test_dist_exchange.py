"""One capture over two RANK PROCESSES (gloo on CPU; the same calls are RCCL send / recv on GPUs): the chunk owners produce the candidate
tables (here with the oracle's candidate decoder standing in for k_viterbi - no GPU in this test), `DescriptorExchange` carries them to the
rank that runs the PRODUCT's sequential FALCON search (HIP-free host logic, tests/native) and carries the accepted-DCI descriptors back;
the descriptors the owners end up with, merged in chunk order, equal the single-process search over the same capture."""
import ctypes as C
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NSF, CHUNK = 24, 4


def _tables(sc, n):
    """per subframe: (tti, cfi, snr, candidate table bytes, cce powers) from the oracle front end + candidate decoder; also the size list"""
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from lsn_testlib import OCell, OracleWorker, TxGen, candidate_table, hosttest, oracle
    h = hosttest()

    class Regs(C.Structure):
        _fields_ = [("nof_regs", C.c_uint32 * 3), ("nof_cce", C.c_uint32 * 3), ("k0", (C.c_uint16 * 800) * 3),
                    ("l", (C.c_uint8 * 800) * 3), ("pcfich_k0", C.c_uint16 * 4), ("ngroups_phich", C.c_uint32)]
    regs = Regs()
    cell = OCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["phich_ng_x6"])
    oracle().o_regs_init.argtypes = [C.POINTER(OCell), C.c_void_p]
    oracle().o_regs_init(C.byref(cell), C.byref(regs))
    cce = (C.c_uint32 * 3)(*regs.nof_cce)
    hs = h.lsnh_search_new(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], cce, 5, 0.99, 0)
    sizes = [h.lsnh_search_size(hs, k) for k in range(h.lsnh_search_nof_sizes(hs))]
    tx = TxGen(**sc)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"], sc["phich_ng_x6"])
    out = []
    for _ in range(n):
        tti, iq, _p = tx.next()
        ow.work(iq, tti)
        cfi = ow.cfi()
        cand, pw = candidate_table(ow.llr(), cce[cfi - 1], sizes, tti % 10)
        out.append((tti, cfi, float(ow.chest().snr_db), np.frombuffer(bytes(cand), dtype=np.uint8).copy(), pw.copy()))
    return h, hs, out


def _pack(rows):
    """a chunk's stage-A results as one byte array: per subframe a 16-byte header (tti, cfi as u32, snr as f32, padding - the candidate table holds
    64-bit fields and stays 8-byte aligned on the wire), table, powers"""
    parts = []
    for tti, cfi, snr, cand, pw in rows:
        parts += [np.array([tti, cfi], np.uint32).view(np.uint8), np.array([snr, 0.0], np.float32).view(np.uint8), cand, pw.view(np.uint8)]
    return np.concatenate(parts)


def _search_chunk(h, hs, blob, nsf):
    per = len(blob) // nsf
    words = []
    for i in range(nsf):
        b = blob[i * per:(i + 1) * per]
        tti, cfi = (int(v) for v in b[:8].view(np.uint32))
        snr = float(b[8:12].view(np.float32)[0])
        cand, pw = b[16:per - 96 * 4].copy(), b[per - 96 * 4:].copy().view(np.float32)
        assert cand.ctypes.data % 8 == 0
        out = (C.c_uint32 * (64 * 6))()
        n = h.lsnh_search_run(hs, tti, cfi, snr, cand.ctypes.data, pw.ctypes.data, 0, out, 64 * 6)
        words.append(np.array([n] + list(out[:6 * n]), np.uint32))
    return np.concatenate(words)


def _worker(rank, world, port, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from ltesniffer_amd import dist as ld
    from lsn_testlib import scenario
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc = scenario("small", seed=31)
    h, hs, rows = _tables(sc, NSF)   # (every rank renders the capture; a rank only USES the subframes of the chunks it owns)
    ex = ld.DescriptorExchange(search_rank=0)
    mine = {}
    for c in range(NSF // CHUNK):
        owner = ex.owner_of(c, world)
        blob = _pack(rows[c * CHUNK:(c + 1) * CHUNK]) if rank == owner else None
        got = ex.tables_up(c, blob)
        grants = _search_chunk(h, hs, got, CHUNK) if rank == 0 else None   # strictly in chunk order on the search rank
        back = ex.grants_down(c, grants)
        if rank == owner:
            mine[c] = np.ascontiguousarray(back).view(np.uint8).view(np.uint32)   # (bytes on the wire)
    dist.barrier()
    q.put((rank, {c: v.tolist() for c, v in mine.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_descriptor_exchange_equals_single_process_search():
    world, port = 2, 31500 + os.getpid() % 2000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = dict(q.get(timeout=600) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    merged = {}
    for r in res.values():
        merged.update(r)
    assert sorted(merged) == list(range(NSF // CHUNK)) and sorted(res[1]) == [1, 3, 5]   # every chunk came back to its owner
    # single process: the same search over the same tables
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    from lsn_testlib import scenario
    h, hs, rows = _tables(scenario("small", seed=31), NSF)
    ref = [_search_chunk(h, hs, _pack(rows[c * CHUNK:(c + 1) * CHUNK]), CHUNK).tolist() for c in range(NSF // CHUNK)]
    assert [merged[c] for c in range(NSF // CHUNK)] == ref
    assert sum(sum(1 for _ in v) for v in ref) > NSF   # DCIs were accepted
