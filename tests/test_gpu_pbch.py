"""GPU parity of the PBCH / MIB decoder (k_pbch_llr + k_pbch_viterbi behind lsn_phy_mib_decode) against the oracle, and file replay
that takes its SFN from the MIB (the reference's DECODE_MIB state)."""
import numpy as np
import pytest

import ltesniffer_amd as la
from lsn_testlib import OracleWorker, TxGen, parse_pcap, scenario
from parity import gen_subframes, gpu_records, oracle_records
from test_pbch_oracle import oracle_mib
from test_file_source import oracle_read, write_capture

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scn,over", [("small", {}), ("cfg1", {}), ("cfg3", dict(dl_min=2, dl_max=3)), ("small", dict(cell_id=301, phich_ng_x6=6, snr_db=6.0)),
                                      ("small", dict(nof_ports=4)), ("cfg2", dict(nof_ports=4, cell_id=77, snr_db=7.0)),
                                      ("small", dict(cp=1)), ("cfg1", dict(cp=1, cell_id=11)), ("small", dict(cp=1, nof_ports=4, nof_prb=50, snr_db=8.0))])
def test_mib_decode_matches_oracle(scn, over):
    sc = scenario(scn, seed=12, start_tti=10 * 1021 + 8, **over)
    tx = TxGen(**sc)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=4)
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], cp=sc.get("cp", 0))
    found = 0
    for _ in range(45):
        tti, iq, _ = tx.next()
        ollr = np.zeros(480, dtype=np.float32)
        r, m = oracle_mib(sc, iq, ollr)
        g, gllr = phy.mib_decode(iq, with_llr=True)
        assert np.array_equal(gllr.view(np.uint32), ollr.view(np.uint32)), (tti, float(np.abs(gllr - ollr).max()))
        assert g["found"] == r
        if r:
            assert (g["sfn"], g["sfn_offset"], g["nof_prb"], g["nof_ports"], g["phich_length"], g["phich_resources_x6"], g["mib_bits"]) == \
                (m.sfn, m.sfn_offset, m.nof_prb, m.nof_ports, m.phich_length, m.phich_ng_x6, m.mib_bits)
            assert tti % 10 == 0 and g["sfn"] == (tti // 10) % 1024
            found += 1
    assert found >= 4
    phy.close()


def test_file_replay_takes_the_sfn_from_the_mib(tmp_path):
    sc = scenario("small", seed=8, start_tti=10 * 700)  # the capture starts at subframe 0 of SFN 700
    tti0, iq, _ = gen_subframes(sc, 40)
    p = str(tmp_path / "cap.cf32")
    write_capture(p, iq, lead=12)
    ow = OracleWorker(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["nof_rx"])
    for i in range(40):
        ow.work(iq[i], tti0 + i, update_meta=1 if i % 20 == 0 else 0)
    orecs = oracle_records(parse_pcap(ow.pcap_bytes()))
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    assert phy.process_file(p, start_tti=la.TTI_FROM_MIB, offset_time=12, update_meta_period=20) == 40
    assert gpu_records(phy) == orecs  # the records carry the TTI, i.e. the SFN found in the MIB
    phy.close()
    # a capture whose first radio frame is unreadable: the replay starts at the first frame with a MIB (10 subframes dropped)
    iq2 = iq.copy()
    iq2[0] = 0
    write_capture(p, iq2, lead=0)
    phy = la.Phy(nof_rx_antennas=sc["nof_rx"], max_batch=8, pcapwriter=la.PcapWriter(None))
    assert phy.setCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"])
    assert phy.process_file(p, start_tti=la.TTI_FROM_MIB) == 30
    recs = parse_pcap(phy.pcapwriter.bytes())
    assert recs and min(r["sfn"] * 10 + r["sf"] for r in recs) >= tti0 + 10 and max(r["sfn"] for r in recs) == 703
    phy.close()
