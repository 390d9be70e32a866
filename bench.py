#!/usr/bin/env python3
"""bench.py - subframes/s of the MI355X-native LTESniffer worker on BASELINE.json's metric config.

A "step" is one pass of the hot path (OFDM -> chest -> PCFICH/PDCCH -> exhaustive Viterbi -> FALCON search ->
PDSCH demod -> turbo -> MAC PDUs) over one resident batch of synthetic subframes.  IQ is already in HBM when the timed
region starts.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), every rank replays its own
synthetic cell (subframes/cells shard with no data-path exchange) -> weak scaling; value = all ranks' subframes / max time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nsf 200] [--config cfg3] [--cpu-sample 120]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import numpy as np
import torch


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nsf", type=int, default=6400, help="subframes per step = length of the resident capture (multiple of --gen)")
    ap.add_argument("--gen", type=int, default=1600, help="distinct synthetic subframes generated (multiple of 20); the capture is this block tiled")
    ap.add_argument("--config", default="cfg3", help="scenario preset: cfg3 = 20 MHz, 150 RNTIs, TM3/TM4 up to 256QAM")
    ap.add_argument("--batch", type=int, default=200, help="subframes per pipeline chunk inside a step")
    ap.add_argument("--cpu-sample", type=int, default=600, help="subframes timed on the CPU oracle (rank 0, N=1 only)")
    ap.add_argument("--sync-steps", action="store_true", help="complete every step (lsn_phy_process_device) before the next one starts")
    ap.add_argument("--no-cpu", action="store_true")
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

    import ltesniffer_amd as la
    from lsn_testlib import scenario
    from parity import gen_subframes, gpu_records, oracle_records, run_oracle

    gen = max(20, (min(args.gen, args.nsf) // 20) * 20)
    nsf = max(gen, (args.nsf // gen) * gen)
    batch = min(args.batch or nsf, nsf)
    from ltesniffer_amd import dist as ld
    sc = scenario(args.config, **ld.rank_workload(args.config, rank))  # one synthetic cell per rank (SURVEY 8d config 5 style)
    tti0, iq, truth = gen_subframes(sc, gen)
    # resident capture [nsf][rx][15*N] interleaved cf32 in HBM: the generated block tiled nsf/gen times (its length is a
    # multiple of 20 subframes, so subframe indices and the SIB pattern stay consistent while the TTI keeps advancing)
    d_iq = torch.from_numpy(iq.view(np.float32)).to(dev).repeat(nsf // gen, 1, 1).contiguous()
    torch.cuda.synchronize()

    pcap = la.PcapWriter(None)  # native MAC-LTE writer (in-memory capture), the reference's pcap-emit surface
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=batch, device=local, pcapwriter=pcap)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    stream = torch.cuda.current_stream().cuda_stream

    def step(i):
        phy.process_device(d_iq.data_ptr(), nsf, tti0 + i * nsf, 500, stream)  # LTESniffer_Core.cc:434: meta update / 500 sf

    # ---- parity gate on the first (cold-state) pass: MAC-LTE record stream vs the CPU oracle on the same subframes ----
    cpu = None
    pcap_diff = None
    ns = min(args.cpu_sample, gen)
    if rank == 0 and not args.no_cpu:
        t = time.perf_counter()
        ow, _, orecs = run_oracle(sc, tti0, iq[:ns], update_meta_period=500, taps=False)
        dt = time.perf_counter() - t
        cpu = {"value": round(ns / dt, 2), "unit": "subframes/s", "cores": 1, "kind": "port",
               "sample": "%d subframes of the same workload (cold RNTI state), scalar C oracle, 1 thread" % ns}
        # the same restatement on many cores: W forked workers, each decoding its own 100-subframe slice with its own (cold) RNTI state -
        # an upper bound for a subframe-parallel CPU run of this code (the sequential RNTI state is not shared), informational only
        try:
            import multiprocessing as mp
            W = max(1, min(16, (os.cpu_count() or 2) // 2))
            per = min(200, gen)

            global _CPU_CTX
            _CPU_CTX = (sc, tti0, iq, gen, per, run_oracle)  # inherited by the forked workers
            with mp.get_context("fork").Pool(W) as pool:
                t = time.perf_counter()
                pool.map(_cpu_slice, range(W))
                dtp = time.perf_counter() - t
            cpu["parallel"] = {"value": round(W * per / dtp, 1), "unit": "subframes/s", "cores": W,
                               "sample": "%d forked workers x %d subframes, independent cold RNTI state each" % (W, per)}
        except Exception as ex:  # the single-core figure above is the contract; this one is optional
            cpu["parallel"] = {"error": str(ex)[:200]}
        chk = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=min(batch, 64), device=local, pcapwriter=la.PcapWriter(None))
        chk.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
        chk.process_host(iq[:ns], tti0, 500)
        g, o = gpu_records(chk), oracle_records(orecs)
        pcap_diff = 0 if g == o else max(1, abs(len(g) - len(o)) + sum(1 for a, b in zip(g, o) if a != b))
        chk.close()

    for i in range(args.warmup):
        step(i)
        pcap.reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    kms = np.zeros(16)
    klaunch = np.zeros(16)
    turbo_bytes = 0
    turbo128_bytes = 0
    algo_bytes = 0
    npdus = 0
    acc = {k: 0 for k in ("nof_tb_decodes", "nof_cb_decodes", "nof_turbo_iterations", "nof_turbo_iterations_run", "nof_ondemand_decodes", "turbo_cyc_rm",
                          "turbo_cyc_map", "turbo_cyc_out", "ms_stage_a", "ms_search", "ms_search_core", "ms_rar", "ms_stage_c", "ms_commit", "ms_wait_front", "ms_wait_slot", "ms_drain")}
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    thr0 = _thread_cpu()
    t0 = time.perf_counter()
    # the K steps are submitted back to back (lsn_phy_submit_device: a step returns once its subframes are searched and queued, its
    # decode / commit tail overlaps the next step's front) and completed by one lsn_phy_wait inside the timed region
    for i in range(args.steps):
        if args.sync_steps:
            step(args.warmup + i)
        else:
            phy.submit_device(d_iq.data_ptr(), nsf, tti0 + (args.warmup + i) * nsf, 500, stream)
        if args.sync_steps or i == args.steps - 1:
            if not args.sync_steps:
                phy.wait()
            p = phy.perf()
            kms += np.array(p.kernel_ms[:])
            klaunch += np.array(p.kernel_launches[:])
            turbo_bytes += p.turbo_algo_bytes
            turbo128_bytes += p.turbo128_algo_bytes
            algo_bytes += p.algo_bytes
            npdus += p.nof_pdus
            for k in acc:
                acc[k] += getattr(p, k)
            if args.sync_steps:
                pcap.reset()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    thr1 = _thread_cpu()
    host_cores_busy = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / dt  # CPU cores this rank kept busy in the timed region
    if world > 1:
        rdev = dev if dist.get_backend() == "nccl" else None
        dt, total = ld.reduce_max_sum(dt, args.steps * nsf, rdev)
        assert total == args.steps * nsf * world
    total_sf = args.steps * nsf * world
    value = total_sf / dt

    if rank == 0:
        per_thr = {}
        for k, v in thr1.items():
            per_thr[k[1]] = per_thr.get(k[1], 0.0) + (v - thr0.get(k, 0.0)) / dt
        busiest = {k: round(v, 2) for k, v in sorted(per_thr.items(), key=lambda kv: -kv[1])[:8] if v >= 0.01}
        dom = int(np.argmax(kms[:len(la.KERNELS)]))
        # roofline of the dominant kernel (one of the two turbo-decoder variants): algorithmic bytes = rate-matched int16
        # LLRs read (E * 2 per code block) + payload bytes written, per launch; duration from HIP events on the launch stream
        k64, k128 = la.KERNELS.index("k_turbo<64>"), la.KERNELS.index("k_turbo<128>")
        kt = k128 if kms[k128] >= kms[k64] else k64
        kbytes = turbo128_bytes if kt == k128 else turbo_bytes - turbo128_bytes
        ach = (kbytes / 1e9) / (kms[kt] / 1e3) if kms[kt] > 0 else 0.0
        # HBM traffic of the same kernel from the PMC counters: collected by SEPARATE rocprofv3 --pmc passes of this command
        # (tools/pmc_summary.py -> profiles/*_pmc_hbm.json); FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md
        traffic, traffic_src = None, None
        try:
            import glob
            cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_hbm.json")))
            cur = os.path.join(ROOT, "profiles", "current.json")
            if os.path.exists(cur) and json.load(open(cur)).get("pmc_hbm"):
                cand = [os.path.join(ROOT, "profiles", json.load(open(cur))["pmc_hbm"])]
            if cand:
                pj = json.load(open(cand[-1])).get(la.KERNELS[kt])
                if pj:
                    traffic = int(pj["fetch_corrected_bytes_per_launch"] + pj["write_bytes_per_launch"])
                    traffic_src = os.path.relpath(cand[-1], ROOT)
        except Exception:
            pass
        # secondary view (the recursions are integer-VALU work, not HBM work): wave-level VALU instructions per launch from a
        # separate rocprofv3 --pmc SQ_INSTS_VALU pass (tools/pmc_valu_summary.py -> profiles/*_pmc_valu.json) over this run's launch time,
        # against the plain-VOP2 issue peak measured on this chip by tools/ubench/valu_rate (profiles/*_valu_ubench.txt)
        valu = None
        try:
            cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_valu.json")))
            if os.path.exists(cur) and json.load(open(cur)).get("pmc_valu"):
                cand = [os.path.join(ROOT, "profiles", json.load(open(cur))["pmc_valu"])]
            if cand:
                vj = json.load(open(cand[-1]))
                pj = vj.get(la.KERNELS[kt])
                if pj and klaunch[kt] > 0 and kms[kt] > 0:
                    g = pj["valu_insts_per_launch"] / (kms[kt] / klaunch[kt] * 1e6)
                    valu = {"wave_insts_per_launch": int(pj["valu_insts_per_launch"]), "achieved_G_per_s": round(g, 1),
                            "peak_G_per_s": vj.get("_peak_G_wave_insts_per_s", 740.0), "frac": round(g / vj.get("_peak_G_wave_insts_per_s", 740.0), 4),
                            "all_kernels_wave_insts_per_subframe": int(sum(v["valu_insts_total"] for k, v in vj.items() if not k.startswith("_") and (k.startswith("k_"))) /
                                                                       max(1, vj.get("_subframes", 3 * 6400))),
                            "source": os.path.relpath(cand[-1], ROOT)}
        except Exception:
            pass
        out = {
            "metric": "subframes/s (20 MHz, 150 RNTIs)", "value": round(value, 1), "unit": "subframes/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32+int16", "data": "synthetic",
            "x_realtime": round(value / 1000.0, 2), "pcap_diff": pcap_diff,
            "config": {"workload": "%s: 20 MHz DL (100 PRB, 2 CRS ports, 2 rx), 150 active RNTIs, TM2/TM3/TM4 mix up to 256QAM, "
                                   "CFI 3, 8-14 DL + 3-6 UL DCIs per subframe (BASELINE.json configs[2])" % args.config
                       if args.config == "cfg3" else args.config,
                       "subframes_per_step": nsf, "steps_pipelined": not args.sync_steps, "distinct_subframes": gen, "gpu_batch": batch, "cells": world, "parallelism": "cell/subframe shards, no collective"},
            "roofline": {"bound": "hbm", "kernel": la.KERNELS[kt], "achieved": round(ach, 2), "peak": 8000.0, "unit": "GB/s",
                         "frac": round(ach / 8000.0, 6), "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": round(kms[kt] / max(1, klaunch[kt]), 4), "launches": int(klaunch[kt]),
                         "algo_bytes_per_launch": int(kbytes / max(1, klaunch[kt])),
                         "dominant_by_time": la.KERNELS[dom], "valu": valu},
            "cpu_baseline": cpu, "host": {"cpu_count": os.cpu_count(), "cores_busy_in_timed_region": round(host_cores_busy, 2),
                                             "busiest_threads": busiest},
            "detail": {"pdus_per_step": npdus / args.steps, "algo_bytes_per_subframe": int(algo_bytes / (args.steps * nsf)),
                       "whole_path_GBps": round(algo_bytes / 1e9 / dt, 2),
                       "per_step": {k: round(v / args.steps, 3) for k, v in acc.items()},
                       "kernel_ms_per_step": {la.KERNELS[k]: round(kms[k] / args.steps, 4) for k in range(len(la.KERNELS))}},
        }
        print(json.dumps(out), flush=True)
    phy.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
