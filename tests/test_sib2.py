"""CPU tests of the SIB2 row: PDSCH_Decoder::decode_SIB + ULSchedule::set_config.
(a) both SIB2 walks (oracle o_rrc.c, product lsn_rrc.cc through the tests/native glue) against the BCCH-DL-SCH messages the reference
itself recorded (tests/golden/pcap_records.json "si_pdus": the UL_MODE capture holds exactly the SystemInformation its decode_SIB
accepted); (b) against an independent UPER encoder with every optional component / extension addition switched on and off;
(c) truncated and random input; (d) the oracle's UL_MODE worker configures itself from the first SIB2 of a synthetic cell and then
decodes the same PUSCH records as a worker that was given the configuration - minus the subframes before the SIB2."""
import json
import os

import numpy as np
import pytest

from lsn_testlib import (REAL_SIB1, REAL_SIB2, SIB2_FIELDS, OracleWorkerUl, encode_sib2, gen_ul_mode_subframes, host_sib2_decode,
                         oracle_sib2_decode, parse_pcap, scenario)

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pcap_records.json")))

# srsENB's stock sib.conf (the eNB behind the reference's example captures)
SRSENB_SIB2 = dict(n_sb=1, hopping_mode=0, pusch_hop_offset=2, enable_64qam=1, group_hopping_enabled=0, group_assignment_pusch=0,
                   sequence_hopping_enabled=0, cyclic_shift=0, root_seq_idx=128, prach_config_idx=3, high_speed_flag=0, zero_corr_zone=5,
                   prach_freq_offset=4)


def test_fixture_constants_are_the_recorded_messages():
    assert FIX["ltesniffer_ul_mode.pcap"]["si_pdus"] == [REAL_SIB2.hex()]
    assert sorted(FIX["ltesniffer_dl_mode.pcap"]["si_pdus"]) == sorted([REAL_SIB2.hex(), REAL_SIB1.hex()])


def test_recorded_sib2_decodes_to_the_enb_configuration():
    r, d, used = oracle_sib2_decode(REAL_SIB2)
    assert r == 2 and d == SRSENB_SIB2
    # the walk ends inside the message, the block behind SIB2 (the message lists two) and the padding follow
    assert 150 < used < 8 * len(REAL_SIB2) - 64
    assert host_sib2_decode(REAL_SIB2) == (2, SRSENB_SIB2)


def test_recorded_sib1_is_not_a_sib2():
    assert oracle_sib2_decode(REAL_SIB1)[0] == 1 and host_sib2_decode(REAL_SIB1)[0] == 1


def _variants():
    rng = np.random.RandomState(7)
    for i in range(300):
        f = dict(n_sb=int(rng.randint(1, 5)), hopping_mode=int(rng.randint(2)), pusch_hop_offset=int(rng.randint(99)), enable_64qam=int(rng.randint(2)),
                 group_hopping_enabled=int(rng.randint(2)), group_assignment_pusch=int(rng.randint(30)), sequence_hopping_enabled=int(rng.randint(2)),
                 cyclic_shift=int(rng.randint(8)), root_seq_idx=int(rng.randint(838)), prach_config_idx=int(rng.randint(64)),
                 high_speed_flag=int(rng.randint(2)), zero_corr_zone=int(rng.randint(16)), prach_freq_offset=int(rng.randint(95)))
        opt = dict(ac_barring=int(rng.randint(3)), group_a=int(rng.randint(2)), srs=int(rng.randint(3)), mbsfn=int(rng.randint(0, 9)) if rng.randint(2) else 0,
                   ul_carrier=int(rng.randint(2)), ul_bw=int(rng.randint(2)), rach_ext=int(rng.randint(2)), rr_ext=int(rng.randint(2)),
                   timers_ext=int(rng.randint(2)), sib_ext=int(rng.randint(2)), nblocks=int(rng.randint(1, 4)), fill=i)
        yield f, opt


def test_encoder_round_trip_with_every_optional_component():
    n_opt = 0
    for f, opt in _variants():
        msg = encode_sib2(**f, **opt)
        r, d, used = oracle_sib2_decode(msg)
        assert r == 2 and d == f, (f, opt, d)
        assert used <= 8 * len(msg) and used > 8 * (len(msg) - 5 * (opt["nblocks"] - 1)) - 8
        assert host_sib2_decode(msg) == (2, f), (f, opt)
        n_opt += sum(1 for k in ("ac_barring", "group_a", "srs", "mbsfn", "rach_ext", "rr_ext", "timers_ext", "sib_ext") if opt[k])
    assert n_opt > 600


def test_encoder_reproduces_the_head_of_the_recorded_message():
    """the independent encoder, fed with the eNB's values, emits the recorded bits of the fields the sniffer reads"""
    msg = encode_sib2(**SRSENB_SIB2)
    # prach-Config .. ul-ReferenceSignalsPUSCH are contiguous: locate them in both messages by decoding and compare the field bits
    def bits(b):
        return "".join("{:08b}".format(x) for x in b)
    want = "{:010b}{:06b}{:01b}{:04b}{:07b}".format(128, 3, 0, 5, 4)
    assert want in bits(msg) and want in bits(REAL_SIB2)
    tail = "{:02b}{:01b}{:07b}{:01b}{:01b}{:05b}{:01b}{:03b}".format(0, 0, 2, 1, 0, 0, 0, 0)
    i, j = bits(msg).index(want), bits(REAL_SIB2).index(want)
    assert bits(msg)[i + 28 + 9:i + 28 + 9 + 21] == tail and bits(REAL_SIB2)[j + 28 + 9:j + 28 + 9 + 21] == tail


def test_truncated_and_random_input_is_rejected_identically():
    rng = np.random.RandomState(3)
    msgs = [encode_sib2(**f, **opt) for f, opt in list(_variants())[:40]] + [REAL_SIB2]
    for m in msgs:
        r_full, _, used = oracle_sib2_decode(m)
        nbytes = (used + 7) // 8
        for cut in range(0, nbytes):
            ro, do, _ = oracle_sib2_decode(m[:cut])
            rh, dh = host_sib2_decode(m[:cut])
            assert ro == rh and do == dh
            assert ro != 2
        assert oracle_sib2_decode(m[:nbytes])[0] == 2
    agree2 = 0
    for i in range(4000):
        m = bytes(rng.randint(0, 256, size=int(rng.randint(1, 60))).astype(np.uint8))
        if i % 2:  # a valid message with a few flipped bits: many stay decodable (with other values), some break
            b = bytearray(msgs[i % len(msgs)])
            for _ in range(int(rng.randint(1, 4))):
                k = int(rng.randint(14, 8 * len(b)))
                b[k >> 3] ^= 0x80 >> (k & 7)
            m = bytes(b)
        ro, do, _ = oracle_sib2_decode(m)
        rh, dh = host_sib2_decode(m)
        assert (ro, do) == (rh, dh)
        agree2 += ro == 2
    assert agree2 > 0
    assert oracle_sib2_decode(b"")[0] == 0 and host_sib2_decode(b"")[0] == 0


def _ul_cell(seed, **over):
    return scenario("cfg2", seed=seed, nof_rx=1, n_rnti=10, dl_min=2, dl_max=3, ul_min=2, ul_max=4, nof_prb=25, mcs_max=18, **over)


def test_oracle_ul_mode_configures_itself_from_sib2():
    """SI messages alternate between the recorded SIB1 and a SIB2 that carries the cell's uplink configuration: the worker without a
    configuration writes nothing until the SIB2, then exactly that SI record, and from the next subframe on the same records as the
    worker that was configured by hand"""
    sc = _ul_cell(21, pusch_hop_offset=4, pct_hop=30)
    sib2 = encode_sib2(cyclic_shift=3, group_assignment_pusch=5, pusch_hop_offset=4, root_seq_idx=22, prach_config_idx=3, zero_corr_zone=1,
                       prach_freq_offset=2, ac_barring=1, srs=1, rr_ext=1, nblocks=2, fill=5)
    nsf = 70
    tti0, iq, sent = gen_ul_mode_subframes(sc, nsf, si_msgs=[REAL_SIB1, sib2])
    given = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], 3, 5, 4)
    auto = OracleWorkerUl(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], None, None)
    assert auto.ul_config() is None
    per_g, per_a, learned_at = [], [], None
    for i in range(nsf):
        per_g.append(given.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 25 == 0 else 0))
        per_a.append(auto.work_ul(iq[i, 0], iq[i, 1], tti0 + i, update_meta=1 if i % 25 == 0 else 0))
        if learned_at is None and auto.ul_config() is not None:
            learned_at = i
    # SI on subframe 5 of even frames, SIB1 first: the SIB2 goes out in frame 2
    assert learned_at == 25 - tti0 % 10 if tti0 % 10 else learned_at == 25
    cfg = auto.ul_config()
    assert cfg["from_sib2"] and (cfg["cyclic_shift"], cfg["delta_ss"], cfg["hopping_offset"]) == (3, 5, 4)
    assert cfg["sib2"]["root_seq_idx"] == 22 and cfg["sib2"]["zero_corr_zone"] == 1 and cfg["sib2"]["prach_freq_offset"] == 2
    assert given.ul_config()["from_sib2"] is False
    assert per_a[:learned_at] == [0] * learned_at and per_a[learned_at] == 1
    ra, rg = parse_pcap(auto.pcap_bytes()), parse_pcap(given.pcap_bytes())
    for r in ra + rg:
        r["tti"] = r["sfn"] * 10 + r["sf"]
    t_learn = (tti0 + learned_at) % 10240
    assert ra[0]["pdu"][:len(sib2)] == sib2 and ra[0]["rnti_type"] == 4 and ra[0]["tti"] == t_learn
    assert not [r for r in ra[1:] if r["tti"] == t_learn]
    # afterwards: the records of the hand-configured worker, except PUSCH grants scheduled before the configuration existed (the 4 / 6 ms
    # schedule) and what the modulation tracking of the other worker had learnt earlier
    key = lambda r: (r["tti"], r["direction"], r["rnti"], r["pdu"])
    late_g = {key(r) for r in rg if 6 < (r["tti"] - t_learn) % 10240 < 5000}
    late_a = [key(r) for r in ra[1:] if (r["tti"] - t_learn) % 10240 > 6]
    assert all(0 < (r["tti"] - t_learn) % 10240 < 5000 for r in ra[1:])
    assert len(late_a) > 20 and sum(1 for k in late_a if k[1] == 0) >= 5  # downlink and uplink records after the self-configuration
    assert len(set(late_a) - late_g) <= 2 and len(late_g - set(late_a)) <= 2
