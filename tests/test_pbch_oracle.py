"""CPU tests of the PBCH / MIB restatement: synthetic eNB (tools/txgen, 36.211 6.6 / 36.212 5.3.1) -> oracle decode loop-back."""
import ctypes as C

import numpy as np
import pytest

from lsn_testlib import OCell, TxGen, oracle, scenario


class OMib(C.Structure):
    _fields_ = [("found", C.c_int), ("sfn", C.c_uint32), ("sfn_offset", C.c_uint32), ("nof_prb", C.c_uint32), ("nof_ports", C.c_uint32),
                ("phich_length", C.c_uint32), ("phich_ng_x6", C.c_uint32), ("mib_bits", C.c_uint32)]


def oracle_mib(sc, iq, llr=None):
    o = oracle()
    o.o_mib_decode_subframe.argtypes = [C.POINTER(OCell), C.c_uint32, C.c_void_p, C.POINTER(OMib), C.c_void_p]
    cell = OCell(sc["nof_prb"], sc["nof_ports"], sc["cell_id"], sc["phich_ng_x6"], 0, sc.get("cp", 0))
    m = OMib()
    iq = np.ascontiguousarray(iq, dtype=np.complex64)
    r = o.o_mib_decode_subframe(C.byref(cell), sc["nof_rx"], iq.ctypes.data, C.byref(m), llr.ctypes.data if llr is not None else None)
    return r, m


@pytest.mark.parametrize("scn,over", [("small", {}), ("cfg1", {}), ("cfg3", dict(dl_min=2, dl_max=3)), ("small", dict(cell_id=301, phich_ng_x6=6, snr_db=8.0)),
                                      ("small", dict(nof_ports=4)), ("small", dict(nof_ports=4, nof_prb=100, cell_id=77, snr_db=8.0)),
                                      ("small", dict(cp=1)), ("cfg1", dict(cp=1, cell_id=11)), ("small", dict(cp=1, nof_ports=4, nof_prb=50, snr_db=8.0))])  # extended CP: 216 symbols, E = 1728
def test_mib_loopback_recovers_sfn_ports_and_bandwidth(scn, over):
    sc = scenario(scn, seed=12, start_tti=10 * 513 + 7, **over)  # starts in the middle of a frame, SFN 513
    tx = TxGen(**sc)
    seen = 0
    for _ in range(64):
        tti, iq, _ = tx.next()
        r, m = oracle_mib(sc, iq)
        if tti % 10 == 0:
            sfn = (tti // 10) % 1024
            assert r == 1 and m.found and m.sfn == sfn and m.sfn_offset == sfn % 4, (tti, m.sfn, m.sfn_offset)
            assert m.nof_prb == sc["nof_prb"] and m.nof_ports == sc["nof_ports"] and m.phich_ng_x6 == sc["phich_ng_x6"] and m.phich_length == 0
            seen += 1
        else:
            assert r == 0, tti  # no PBCH in the other subframes
    assert seen >= 6


def test_mib_sfn_wraps_at_1024():
    sc = scenario("small", seed=3, start_tti=10 * 1023)
    tx = TxGen(**sc)
    got = []
    for _ in range(21):
        tti, iq, _ = tx.next()
        if tti % 10 == 0:
            got.append(oracle_mib(sc, iq)[1].sfn)
    assert got == [1023, 0, 1]
