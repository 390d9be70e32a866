#!/usr/bin/env python3
"""bench.py - subframes/s of the MI355X-native LTESniffer worker on BASELINE.json's metric config.

A "step" is one pass of the hot path (OFDM -> chest -> PCFICH/PDCCH -> exhaustive Viterbi -> FALCON search -> PDSCH demod ->
rate de-matching -> turbo -> MAC PDUs into the MAC-LTE pcap writer) over `--reps` replays of a resident capture of `--nsf` synthetic
subframes (the TTI keeps advancing, the sequential RNTI / MCS-table state carries across replays).  IQ is in HBM before the timed
region starts.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nsf 6400] [--reps 10] [--config cfg3] [--cpu-sample 1600]

N > 1: `--gpus N` launches N ranks itself (python -m torch.distributed.run, one process per GPU, backend nccl = RCCL) unless it already
runs under one (RANK / WORLD_SIZE set, which is how the driver starts it).  Every rank replays its own synthetic cell (BASELINE
configs[4]: cells shard with no data-path exchange) -> weak scaling; value = all ranks' subframes / max-over-ranks time.

Parity gate (rank 0): `pcap_diff` describes the TIMED stream.  (1) the first `--cpu-sample` subframes of the capture from cold state:
record stream == the CPU oracle's, record by record; (2) the records of the K timed steps (pipelined lsn_phy_submit_device, chunk size
--batch) hash to the same digest as a second Phy that walks the same W + K steps synchronously (lsn_phy_process_device, another chunk
size) - the stream the oracle comparison anchors."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def _maybe_spawn(argv):
    """--gpus N without a launcher: start N ranks (one per GPU) and relay rank 0's line"""
    n = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            n = int(argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    sys.exit(subprocess.call(cmd, env=env))


_maybe_spawn(sys.argv[1:])

import numpy as np  # noqa: E402
import torch  # noqa: E402

_CPU_CTX = None


def _thread_cpu():
    """CPU seconds (user+system) of every thread of this process, keyed by (tid, name)"""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open(f"/proc/self/task/{tid}/stat").read()
        except OSError:
            continue
        name = st[st.index("(") + 1:st.rindex(")")]
        f = st[st.rindex(")") + 2:].split()
        out[(int(tid), name)] = (int(f[11]) + int(f[12])) / tck
    return out


def _cpu_slice(k):
    sc, tti0, iq, gen, per, run_oracle = _CPU_CTX
    a = (k * per) % max(1, gen - per)
    run_oracle(sc, tti0 + a, iq[a:a + per], update_meta_period=500, taps=False)
    return 0


def _profile_json(key):
    """profiles/current.json names the rocprofv3 summaries of this tree (tools/gpu_profile.sh); None when absent"""
    try:
        cur = json.load(open(os.path.join(ROOT, "profiles", "current.json")))
        return json.load(open(os.path.join(ROOT, "profiles", cur[key]))), cur[key]
    except Exception:
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nsf", type=int, default=6400, help="length of the resident capture in subframes (multiple of --gen)")
    ap.add_argument("--reps", type=int, default=10, help="replays of the resident capture per step (a step = nsf * reps subframes)")
    ap.add_argument("--gen", type=int, default=1600, help="distinct synthetic subframes generated (multiple of 20); the capture is this block tiled")
    ap.add_argument("--config", default="cfg3", help="scenario preset: cfg3 = 20 MHz, 150 RNTIs, TM3/TM4 up to 256QAM")
    ap.add_argument("--batch", type=int, default=800, help="subframes per pipeline chunk inside a submit")
    ap.add_argument("--cpu-sample", type=int, default=1600, help="subframes of the capture decoded by the CPU oracle from cold state (rank 0): parity gate + cpu_baseline")
    ap.add_argument("--no-cpu", action="store_true", help="skip the oracle leg (parity gate part 1 and cpu_baseline)")
    ap.add_argument("--no-check", action="store_true", help="skip the synchronous second pass (parity gate part 2)")
    ap.add_argument("--shard", choices=("cells", "capture"), default="cells",
                    help="N > 1: 'cells' = one synthetic cell per rank (weak scaling, the default and what BASELINE configs[4] asks for); 'capture' = ONE capture "
                         "whose chunks go round-robin to the N GPUs (lsn_phy_create_multi on rank 0; the other ranks only hold their GPU) - strong scaling, "
                         "bounded by the sequential FALCON search on one host thread")
    ap.add_argument("--no-legs", action="store_true", help="skip the PCIe-inclusive legs (host buffers, capture file)")
    ap.add_argument("--leg-nsf", type=int, default=12800, help="subframes of the capture file / host buffer of the PCIe-inclusive legs")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LSN_DIST_BACKEND", "nccl")  # nccl = RCCL on ROCm; "gloo" only to smoke-test N > 1 on a 1-GPU box
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
            local = local % max(1, torch.cuda.device_count())
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the library has no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # host threads and host buffers next to the GPU: pinned staging, the page-cache pages of the capture file and the caller's own arrays are
    # first touched by this process, and a far-socket source halves the PCIe rate (the engine pins its own threads the same way)
    try:
        pr = torch.cuda.get_device_properties(local)
        bus = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bus).read())
        if node >= 0:
            cpus = set()
            for tok in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
                a, _, b = tok.partition("-")
                cpus.update(range(int(a), int(b or a) + 1))
            cpus &= os.sched_getaffinity(0)
            if cpus:
                os.sched_setaffinity(0, cpus)
    except Exception:
        pass

    # N ranks share the host: a rank keeps ~3 cores busy with 8 decode threads (host.cores_busy_in_timed_region).  Under a CPU quota smaller
    # than that (cgroup cpu.max; the 1-GPU boxes of this pool give 16 cores) the CFS throttle stalls whole ranks, so the pipeline is narrowed
    # instead: fewer decode threads per rank (each costs ~0.3 core, search + front + commit + writer ~0.9).
    host_quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        host_quota = float(q) / float(per) if q != "max" else float(len(os.sched_getaffinity(0)))
    except Exception:
        host_quota = float(len(os.sched_getaffinity(0)))
    if world > 1 and "LSN_DECODE_THREADS" not in os.environ and host_quota / world < 3.4:
        os.environ["LSN_DECODE_THREADS"] = str(max(3, min(8, int((host_quota / world - 0.9) / 0.3))))

    import ltesniffer_amd as la
    from lsn_testlib import scenario
    from parity import gen_subframes, gpu_records, oracle_records, run_oracle
    from ltesniffer_amd import dist as ld

    gen = max(20, (min(args.gen, args.nsf) // 20) * 20)
    nsf = max(gen, (args.nsf // gen) * gen)
    reps = max(1, args.reps)
    batch = min(args.batch or nsf, nsf)
    sc = scenario(args.config, **ld.rank_workload(args.config, rank))  # one synthetic cell per rank
    tti0, iq, _ = gen_subframes(sc, gen)
    # resident capture [nsf][rx][15*N] interleaved cf32 in HBM: the generated block tiled nsf/gen times (its length is a multiple of 20
    # subframes, so subframe indices and the SIB pattern stay consistent while the TTI keeps advancing)
    d_iq = torch.from_numpy(iq.view(np.float32)).to(dev).repeat(nsf // gen, 1, 1).contiguous()
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    sf_per_step = nsf * reps

    capture_mode = args.shard == "capture" and (world > 1 or os.environ.get("LSN_BENCH_DEVICES"))
    devices = None
    if capture_mode:
        devices = [int(x) for x in os.environ["LSN_BENCH_DEVICES"].split(",")] if os.environ.get("LSN_BENCH_DEVICES") else list(range(world))
    if capture_mode and rank != 0:  # the capture is driven by rank 0's process (one search thread, one record stream); this rank's GPU is one of its devices
        dist.barrier()
        dist.barrier()
        dist.destroy_process_group()
        return
    pcap = la.PcapWriter(None)  # native MAC-LTE writer, the reference's pcap-emit surface; the timed stream is digested, not kept
    pcap.set_store(False)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, device=local, pcapwriter=pcap, devices=devices)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])

    def submit_step(p, i, sync=False):
        for r in range(reps):
            t = (tti0 + (i * reps + r) * nsf) % 10240
            if sync:
                p.process_device(d_iq.data_ptr(), nsf, t, 500, stream)  # LTESniffer_Core.cc:434: meta-format update every 500 subframes
            else:
                p.submit_device(d_iq.data_ptr(), nsf, t, 500, stream)

    for i in range(args.warmup):
        submit_step(phy, i)
    phy.wait()
    pcap.reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    thr0 = _thread_cpu()
    t0 = time.perf_counter()
    # the K steps are submitted back to back (lsn_phy_submit_device: a submit returns once its subframes are searched and queued, the
    # decode / commit tail overlaps the next submit) and completed by one lsn_phy_wait inside the timed region
    for i in range(args.steps):
        submit_step(phy, args.warmup + i)
    phy.wait()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    thr1 = _thread_cpu()
    p = phy.perf()
    timed_digest, timed_bytes = pcap.digest()
    timed_records = pcap.nof_records()
    host_cores_busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / dt
    if world > 1 and not capture_mode:
        rdev = dev if dist.get_backend() == "nccl" else None
        dt, total = ld.reduce_max_sum(dt, args.steps * sf_per_step, rdev)
        assert total == args.steps * sf_per_step * world
    total_sf = args.steps * sf_per_step * (1 if capture_mode else world)
    value = total_sf / dt
    phy.close()

    # ---------------------------------------------------------------- parity gate + CPU baseline (rank 0, outside the timed region)
    cpu, parity, pcap_diff = None, None, None
    if rank == 0:
        parity = {"timed_records": timed_records, "timed_bytes": timed_bytes, "timed_digest": "%016x" % timed_digest}
        ns = max(20, min(args.cpu_sample, gen))
        orecs = None
        if not args.no_cpu:
            t = time.perf_counter()
            _, _, orecs = run_oracle(sc, tti0, iq[:ns], update_meta_period=500, taps=False)
            dto = time.perf_counter() - t
            cpu = {"value": round(ns / dto, 2), "unit": "subframes/s", "cores": 1, "kind": "port",
                   "sample": "the first %d subframes of the same capture (cold RNTI state), scalar C oracle, 1 thread" % ns}
            if world == 1:
                # the same restatement on many cores: forked workers on independent 200-subframe slices, each with its own (cold) state - an
                # upper bound for a subframe-parallel CPU run of this code (the sequential RNTI state is not shared), informational only
                try:
                    import multiprocessing as mp
                    W = max(1, min(16, (os.cpu_count() or 2) // 2))
                    per = min(200, gen)
                    global _CPU_CTX
                    _CPU_CTX = (sc, tti0, iq, gen, per, run_oracle)
                    with mp.get_context("fork").Pool(W) as pool:
                        t = time.perf_counter()
                        pool.map(_cpu_slice, range(W))
                        dtp = time.perf_counter() - t
                    cpu["parallel"] = {"value": round(W * per / dtp, 1), "unit": "subframes/s", "cores": W,
                                       "sample": "%d forked workers x %d subframes, independent cold RNTI state each" % (W, per)}
                except Exception as ex:
                    cpu["parallel"] = {"error": str(ex)[:200]}
        if not args.no_check:
            cw = la.PcapWriter(None)
            chk = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=96, device=local, pcapwriter=cw)
            chk.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
            # the walk below is the SAME subframe sequence the timed Phy saw (W + K steps from cold state), cut differently: first the
            # oracle's sample as a call of its own (records kept and compared one by one), then the rest with only the digest kept
            chk.process_device(d_iq.data_ptr(), ns, tti0 % 10240, 500, stream)
            diff_oracle = None
            if orecs is not None:
                g, o = gpu_records(chk), oracle_records(orecs)
                diff_oracle = 0 if g == o else max(1, abs(len(g) - len(o)) + sum(1 for a, b in zip(g, o) if a != b))
                parity.update({"oracle_subframes": ns, "oracle_records": len(o), "oracle_mismatches": diff_oracle})
            cw.set_store(False)
            stride = d_iq[0].numel() * 4
            first = True
            for i in range(args.warmup + args.steps):
                if i == args.warmup:
                    cw.reset()
                for r in range(reps):
                    t = (tti0 + (i * reps + r) * nsf) % 10240
                    if first:  # the part of the first replay behind the oracle's sample
                        first = False
                        if nsf > ns:
                            chk.process_device(d_iq.data_ptr() + ns * stride, nsf - ns, (t + ns) % 10240, 500, stream)
                    else:
                        chk.process_device(d_iq.data_ptr(), nsf, t, 500, stream)
            if args.warmup == 0:
                parity["note"] = "warmup 0: the synchronous pass kept the oracle's sample out of its digest; digests are not comparable"
            sd, sb = cw.digest()
            same = (sd == timed_digest and sb == timed_bytes and cw.nof_records() == timed_records) if args.warmup > 0 else None
            parity.update({"sync_records": cw.nof_records(), "sync_digest": "%016x" % sd, "timed_equals_sync": same})
            chk.close()
            if diff_oracle is not None and same is not None:
                pcap_diff = diff_oracle + (0 if same else max(1, abs(cw.nof_records() - timed_records)))

    # ---------------------------------------------------------------- PCIe-inclusive legs ("first H2D -> last PDU on host")
    # Never `value`: the same capture (a) handed over as host buffers (lsn_phy_process_host: PCIe copies overlapped with the pipeline) and
    # (b) replayed from a cf32 file in the page cache (lsn_phy_process_file, the reference's file mode, LTESniffer_Core.cc:240-262,365).
    legs = None
    if rank == 0 and world == 1 and not args.no_legs:
        legs = {}
        try:
            ln = max(nsf, (args.leg_nsf // nsf) * nsf)
            sf_bytes = iq[0].nbytes
            host = torch.from_numpy(np.tile(iq, (ln // gen, 1, 1))).pin_memory()
            lp = la.PcapWriter(None)
            lp.set_store(False)
            lphy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, device=local, pcapwriter=lp)
            lphy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
            for name, arr in (("host_pinned", host.numpy()), ("host_pageable", np.array(host.numpy()[:nsf]))):
                lphy.process_host(arr, tti0 % 10240, 500)  # warm state / first-touch
                t = time.perf_counter()
                lphy.process_host(arr, tti0 % 10240, 500)
                dtl = time.perf_counter() - t
                legs[name] = {"subframes_per_s": round(arr.shape[0] / dtl, 1), "GB_per_s_over_pcie": round(arr.shape[0] * sf_bytes / dtl / 1e9, 2),
                              "subframes": int(arr.shape[0]), "pcie_gen5_x16_GB_per_s": 63.0}
            path = "/dev/shm/lsn_bench_capture_%d.cf32" % os.getpid()
            block = np.ascontiguousarray(np.transpose(iq, (0, 2, 1)))  # file mode: antennas interleaved per sample
            with open(path, "wb") as f:
                for _ in range(ln // gen):
                    block.tofile(f)
            try:
                lphy.process_file(path, start_tti=tti0 % 10240, update_meta_period=500)  # warm page cache + state
                t = time.perf_counter()
                done = lphy.process_file(path, start_tti=tti0 % 10240, update_meta_period=500)
                dtl = time.perf_counter() - t
                legs["file_replay"] = {"subframes_per_s": round(done / dtl, 1), "GB_per_s_from_file": round(done * sf_bytes / dtl / 1e9, 2), "subframes": int(done),
                                       "x_realtime": round(done / dtl / 1000.0, 1), "storage": "tmpfs (/dev/shm) = page cache; pread threads -> pinned blocks -> PCIe"}
            finally:
                os.remove(path)
            lphy.close()
            del host
            # cold state: a fresh engine (empty RNTI histograms, no MCS-table knowledge, default meta formats) on the resident capture - the
            # start of a replay, where unknown-table grants are decoded twice and the search walks every format
            cw0 = la.PcapWriter(None)
            cw0.set_store(False)
            cphy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, device=local, pcapwriter=cw0)
            cphy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
            torch.cuda.synchronize()
            t = time.perf_counter()
            cphy.submit_device(d_iq.data_ptr(), nsf, tti0 % 10240, 500)
            cphy.wait()
            dtl = time.perf_counter() - t
            cp = cphy.perf()
            legs["cold_state_first_pass"] = {"subframes_per_s": round(nsf / dtl, 1), "subframes": int(nsf), "tb_decodes_per_subframe": round(cp.nof_tb_decodes / nsf, 2),
                                             "turbo_iterations_per_subframe": round(cp.nof_turbo_iterations / nsf, 1),
                                             "note": "includes pipeline fill and drain of one %d-subframe block" % nsf}
            cphy.close()
        except Exception as ex:
            legs["error"] = str(ex)[:300]

    # ---------------------------------------------------------------- the other BASELINE.json configurations, one short resident pass each
    # (configs[1]: 20 MHz, 32 RNTIs, TM2 64QAM; configs[3]: 20 MHz UL+DL, 64 RNTIs, PUSCH at n + 4 with 16/64QAM turbo decodes) - context
    # numbers next to the headline, never `value`; their parity is covered by tests/test_gpu_parity.py and tests/test_gpu_ul.py
    if legs is not None and "error" not in legs:
        try:
            sc2 = scenario("cfg2", seed=2)
            t2, iq2, _ = gen_subframes(sc2, 400)
            d2 = torch.from_numpy(iq2.view(np.float32)).to(dev).repeat(8, 1, 1).contiguous()
            w2 = la.PcapWriter(None)
            w2.set_store(False)
            p2 = la.Phy(nof_rx_antennas=sc2["nof_rx"], max_batch=batch, device=local, pcapwriter=w2)
            p2.setCell(sc2["nof_prb"], sc2["nof_ports"], sc2["cell_id"])
            n2 = d2.shape[0]
            p2.process_device(d2.data_ptr(), n2, t2 % 10240, 500, stream)
            t = time.perf_counter()
            for r in range(4):
                p2.submit_device(d2.data_ptr(), n2, (t2 + (r + 1) * n2) % 10240, 500, stream)
            p2.wait()
            dtl = time.perf_counter() - t
            legs["cfg2_32_rnti_tm2_64qam"] = {"subframes_per_s": round(4 * n2 / dtl, 1), "subframes": 4 * n2, "records": w2.nof_records()}
            p2.close()
            del d2
            from lsn_testlib import gen_ul_mode_subframes
            sc4 = scenario("cfg2", seed=4, nof_rx=1, n_rnti=64, ul_min=2, ul_max=4, mcs_min=0, mcs_max=28, snr_db=28.0)
            t4, iq4, sent4 = gen_ul_mode_subframes(sc4, 200, ul_snr_db=22.0)
            h4 = torch.from_numpy(np.tile(iq4, (16, 1, 1))).pin_memory()
            w4 = la.PcapWriter(None)
            w4.set_store(False)
            p4 = la.Phy(nof_rx_antennas=2, sniffer_mode=1, max_batch=200, device=local, pcapwriter=w4)
            p4.setCell(sc4["nof_prb"], sc4["nof_ports"], sc4["cell_id"])
            p4.setUlConfig(3, 5)
            p4.process_host(h4.numpy(), t4 % 10240, 500)
            w4.reset()
            t = time.perf_counter()
            p4.process_host(h4.numpy(), t4 % 10240, 500)
            dtl = time.perf_counter() - t
            legs["cfg4_ul_mode_64_rnti"] = {"subframes_per_s": round(h4.shape[0] / dtl, 1), "subframes": int(h4.shape[0]), "records": w4.nof_records(),
                                            "pusch_sent_per_subframe": round(len(sent4) / 200.0, 2), "input": "host buffers (two antenna streams), PCIe included"}
            p4.close()
        except Exception as ex:
            legs["other_configs_error"] = str(ex)[:300]

    if rank == 0:
        kms = np.array(p.kernel_ms[:])
        klaunch = np.array(p.kernel_launches[:])
        per_thr = {}
        for k, v in thr1.items():
            per_thr[k[1]] = per_thr.get(k[1], 0.0) + (v - thr0.get(k, 0.0)) / dt
        busiest = {k: round(v, 2) for k, v in sorted(per_thr.items(), key=lambda kv: -kv[1])[:8] if v >= 0.01}
        nk = len(la.KERNELS)
        dom = int(np.argmax(kms[:nk]))
        sf_rank = args.steps * sf_per_step  # subframes this rank processed in the timed region
        # roofline of the dominant kernel (one of the two turbo-decoder variants).  Algorithmic bytes = what the decoder must move per code
        # block: its K + 12 packed soft words (4 B each, written by k_rm) in, payload bytes out.  Duration: HIP events on the launch stream.
        k64, k128 = la.KERNELS.index("k_turbo<64>"), la.KERNELS.index("k_turbo<128>")
        kt = k128 if kms[k128] >= kms[k64] else k64
        kbytes = p.turbo128_algo_bytes if kt == k128 else p.turbo_algo_bytes - p.turbo128_algo_bytes
        ach = (kbytes / 1e9) / (kms[kt] / 1e3) if kms[kt] > 0 else 0.0
        # HBM traffic of the same kernel from the PMC counters: SEPARATE rocprofv3 --pmc passes of this command (tools/gpu_profile.sh ->
        # profiles/*_pmc_hbm.json), FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md; per launch like `achieved`
        traffic, traffic_src = None, None
        hj, hname = _profile_json("pmc_hbm")
        if hj and hj.get(la.KERNELS[kt]):
            e = hj[la.KERNELS[kt]]
            traffic = int(e["fetch_corrected_bytes_per_launch"] + e["write_bytes_per_launch"])
            traffic_src = "profiles/" + hname
        # secondary view (the recursions are integer-VALU work): wave-level VALU instructions per SUBFRAME from a separate
        # --pmc pass (a property of the workload), over this run's kernel time per subframe = instruction rate while the kernel is resident;
        # peak = 256 CUs x 4 SIMDs x 2.4 GHz / cycles per wave64 VALU instruction as measured by tools/ubench/valu_rate on this chip
        valu = None
        vj, vname = _profile_json("pmc_sq")
        try:
            if vj and vj.get(la.KERNELS[kt]) and kms[kt] > 0:
                sub = float(vj["_subframes"])
                peak = float(vj.get("_peak_G_wave_insts_per_s", 1228.8))
                ins = vj[la.KERNELS[kt]]["SQ_INSTS_VALU"]["total"] / sub
                g = ins / (kms[kt] / sf_rank * 1e6)
                allk = sum(v["SQ_INSTS_VALU"]["total"] for k, v in vj.items() if not k.startswith("_") and "SQ_INSTS_VALU" in v and k.startswith("k_")) / sub
                valu = {"kernel_wave_insts_per_subframe": int(ins), "achieved_G_per_s": round(g, 1), "peak_G_per_s": peak, "frac": round(g / peak, 4),
                        "all_kernels_wave_insts_per_subframe": int(allk), "chip_G_per_s_at_this_rate": round(allk * value / world / 1e9, 1),
                        "chip_frac": round(allk * value / world / 1e9 / peak, 4), "source": "profiles/" + vname}
        except Exception:
            valu = None
        out = {
            "metric": "subframes/s (20 MHz, 150 RNTIs)", "value": round(value, 1), "unit": "subframes/s", "n_gpus": world if not capture_mode or world > 1 else len(devices),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if capture_mode else "weak", "vs_baseline": None, "dtype": "f32+int16", "data": "synthetic",
            "x_realtime": round(value / 1000.0 / (1 if capture_mode else world), 2), "pcap_diff": pcap_diff, "parity": parity,
            "config": {"workload": "%s: 20 MHz DL (100 PRB, 2 CRS ports, 2 rx), 150 active RNTIs, TM2/TM3/TM4 mix up to 256QAM, "
                                   "CFI 3, 8-14 DL + 3-6 UL DCIs per subframe (BASELINE.json configs[2])" % args.config
                       if args.config == "cfg3" else args.config,
                       "subframes_per_step": sf_per_step, "resident_capture_subframes": nsf, "replays_per_step": reps, "steps_pipelined": True,
                       "distinct_subframes": gen, "gpu_batch": batch, "cells": 1 if capture_mode else world,
                       "parallelism": ("one capture, chunks round-robin over devices %s, shared sequential search" % devices) if capture_mode else "one cell per GPU, no collective"},
            "roofline": {"bound": "hbm", "kernel": la.KERNELS[kt], "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(ach / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": round(kms[kt] / max(1, klaunch[kt]), 4), "launches": int(klaunch[kt]),
                         "algo_bytes_per_launch": int(kbytes / max(1, klaunch[kt])),
                         "dominant_by_time": la.KERNELS[dom], "valu": valu},
            "legs": legs, "cpu_baseline": cpu, "host": {"cpu_count": os.cpu_count(), "cpu_quota_cores": host_quota, "decode_threads": int(os.environ.get("LSN_DECODE_THREADS", "8")), "cores_busy_in_timed_region": round(host_cores_busy, 2),
                                             "busiest_threads": busiest},
            "detail": {"pdus_per_subframe": round(p.nof_pdus / sf_rank, 3), "algo_bytes_per_subframe": int(p.algo_bytes / sf_rank),
                       "whole_path_GBps": round(p.algo_bytes * (1 if capture_mode else world) / 1e9 / dt, 2), "timed_region_s": round(dt, 3),
                       "per_6400_subframes": {k: round(getattr(p, k) * 6400.0 / sf_rank, 3) for k in
                                              ("nof_tb_decodes", "nof_cb_decodes", "nof_turbo_iterations", "nof_ondemand_decodes", "turbo_cyc_rm", "turbo_cyc_map",
                                               "turbo_cyc_out", "ms_ondemand_commit", "ms_stage_a", "ms_search", "ms_search_core", "ms_rar", "ms_stage_c", "ms_commit", "ms_wait_front",
                                               "ms_wait_slot", "ms_drain")},
                       "ondemand_at_commit_per_6400": [round(p.nof_ondemand_commit[k] * 6400.0 / sf_rank, 2) for k in range(4)],
                       "kernel_ms_per_6400_subframes": {la.KERNELS[k]: round(kms[k] * 6400.0 / sf_rank, 4) for k in range(nk)}},
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
