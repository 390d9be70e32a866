"""Driver of oracle/_ref/libref_falcon_grant.so: the REFERENCE'S OWN lib/src/phy/falcon_phch/ul_sniffer_pusch.c (DCI 0 -> PUSCH grant: PRB allocation with
type-1 hopping, both uplink MCS tables, row 32A) and dl_sniffer_pdsch.c (MIMO configuration, transport-block enabling, TBS of SI / P / RA-RNTI grants)
compiled from /root/reference (oracle/Makefile.ref; stand-in srsRAN types, oracle/ref_shim_search/grant_glue.c), with the oracle's TBS table bound in.
Sweeps of unpacked DCI fields go through the reference and through the oracle's restatement (o_dci.c).  Test infrastructure only."""
import ctypes as C
import hashlib
import os

from lsn_testlib import OCell, oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_falcon_grant.so")
REF_SOURCES = ["lib/src/phy/falcon_phch/ul_sniffer_pusch.c", "lib/src/phy/falcon_phch/dl_sniffer_pdsch.c"]


class ODciUl(C.Structure):  # o_dci_ul_t
    _fields_ = [("rnti", C.c_uint16), ("L", C.c_uint32), ("ncce", C.c_uint32), ("freq_hop_fl", C.c_uint32), ("riv", C.c_uint32), ("mcs_idx", C.c_uint32),
                ("rv", C.c_int), ("ndi", C.c_uint32), ("tpc", C.c_uint32), ("n_dmrs", C.c_uint32), ("cqi_req", C.c_uint32), ("hop_type", C.c_int)]


class OPuschGrant(C.Structure):  # o_pusch_grant_t
    _fields_ = [("L_prb", C.c_uint32), ("n_prb", C.c_uint32), ("mcs_idx", C.c_uint32), ("mod", C.c_int), ("tbs", C.c_int), ("rv", C.c_int),
                ("n_prb2", C.c_uint32), ("hop", C.c_uint32)]


def ul_sweep():
    """(nof_prb, cp, n_rb_ho, table_256, riv, hop, mcs, cqi) tuples: every resource indication value of six bandwidths x hopping kinds x offsets on a
    thinned MCS grid, and every MCS index on a thinned allocation grid"""
    for nprb in (6, 15, 25, 50, 75, 100):
        nriv = nprb * (nprb + 1) // 2
        for ho in (0, 3, 8, 20) if nprb >= 25 else (0, 2):
            for hop in (-1, 0, 1, 2, 3):
                for riv in range(0, nriv + 3):
                    for mcs in (0, 10, 20, 28):
                        yield nprb, 0, ho, (riv + mcs) & 1, riv, hop, mcs, 0
                for riv in range(0, nriv, 7):
                    for mcs in range(32):
                        for t in (0, 1):
                            yield nprb, riv & 1, ho, t, riv, hop, mcs, (riv >> 1) & 1


class Reference:
    name = "reference"

    def __init__(self):
        self.lib = C.CDLL(REF_SO)
        o = oracle()
        self.lib.ref_grant_bind.argtypes = [C.c_void_p]
        self.lib.ref_grant_bind(C.cast(o.o_tbs_from_idx, C.c_void_p))
        self.lib.ref_ul_grant.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        self.lib.ref_config_mimo.argtypes = [C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_void_p]
        self.lib.ref_compute_tb_common.argtypes = [C.c_int, C.c_uint16, C.c_uint32, C.c_int, C.c_void_p]

    def ul(self, nprb, cp, ho, t256, riv, hop, mcs, cqi):
        """-> None (the reference refuses the grant) | (L_prb, n_prb slot 0, n_prb slot 1, hopping kind, modulation bits, tbs, rv, nof_re)"""
        out = (C.c_int32 * 9)()
        rc = self.lib.ref_ul_grant(nprb, cp, ho, t256, riv, hop, mcs, 0, cqi, out)
        if rc != 0:
            return None
        return (out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[8])

    def mimo(self, ports, fmt, pinfo, nof_tb):
        out = (C.c_int32 * 3)()
        rc = self.lib.ref_config_mimo(ports, fmt, pinfo, nof_tb, out)
        return (rc,) + (tuple(out) if rc == 0 else ())

    def tb_common(self, fmt, rnti, mcs, is2):
        out = (C.c_int32 * 5)()
        rc = self.lib.ref_compute_tb_common(fmt, rnti, mcs, is2, out)
        return (rc,) + (tuple(out) if rc == 0 else ())


class Oracle:
    name = "oracle"

    def __init__(self):
        self.o = oracle()
        self.o.o_ra_ul_dci_to_grant.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.o.o_ra_ul_dci_to_grant_256.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.o.o_tbs_from_idx.restype = C.c_int
        self.o.o_tbs_from_idx.argtypes = [C.c_int, C.c_uint32]

    def ul(self, nprb, cp, ho, t256, riv, hop, mcs, cqi):
        cell = OCell(nprb, 2, 1, 1, ho, cp)
        d = ODciUl(0x100, 0, 0, 1 if hop >= 0 else 0, riv, mcs, 0, 0, 0, 0, cqi, hop)
        g = OPuschGrant()
        rc = (self.o.o_ra_ul_dci_to_grant_256 if t256 else self.o.o_ra_ul_dci_to_grant)(C.byref(cell), C.byref(d), C.byref(g))
        if rc != 0:
            return None
        nsym = 2 * ((6 if cp else 7) - 1)
        return (g.L_prb, g.n_prb, g.n_prb2, g.hop, g.mod, g.tbs, g.rv, nsym * g.L_prb * 12)


class ODciDl(C.Structure):  # o_dci_dl_t
    class TB(C.Structure):
        _fields_ = [("mcs_idx", C.c_uint32), ("rv", C.c_int), ("ndi", C.c_uint32), ("cw_idx", C.c_uint32)]
    _fields_ = [("rnti", C.c_uint16), ("format", C.c_int), ("L", C.c_uint32), ("ncce", C.c_uint32), ("alloc_type", C.c_int), ("rbg_bitmask", C.c_uint32),
                ("t1_vrb_bitmask", C.c_uint32), ("t1_rbg_subset", C.c_uint32), ("t1_shift", C.c_uint32), ("riv", C.c_uint32), ("t2_dist", C.c_int),
                ("t2_ngap2", C.c_int), ("t2_nprb1a_is2", C.c_int), ("pid", C.c_uint32), ("tb", TB * 2), ("tb_cw_swap", C.c_uint32), ("pinfo", C.c_uint32),
                ("tpc", C.c_uint32), ("is_ra_order", C.c_int)]


def mimo_sweep():
    """(ports, format, precoding information, enabled transport blocks): every format of srsran_dci_format_t the search can hand over"""
    for ports in (1, 2, 4):
        for fmt in range(9):
            for pinfo in range(8):
                for nof_tb in (0, 1, 2, 3):
                    yield ports, fmt, pinfo, nof_tb


def tb_common_sweep():
    """(format, rnti, mcs index, n_prb1a is 2): SI / P / RA-RNTI grants in formats 1A and 1C, and a format they may not use"""
    for rnti in (0xFFFF, 0xFFFE, 0x0001, 0x0005, 0x000A):
        for fmt in (2, 4, 1):
            for mcs in range(32):
                for is2 in (0, 1):
                    yield fmt, rnti, mcs, is2


def oracle_mimo(o, ports, fmt, pinfo, nof_tb):
    from lsn_testlib import OGrant
    cell = OCell(50, ports, 1, 1, 0, 0)
    d, g = ODciDl(), OGrant()
    d.pinfo, g.nof_tb = pinfo, nof_tb
    o.o_config_mimo.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    rc = o.o_config_mimo(C.byref(cell), fmt, C.byref(d), C.byref(g))
    return (-rc,) + ((g.tx_scheme, g.pmi, g.nof_layers) if rc == 0 else ())  # error classes 1 / 2 / 3 = DL_SNIFFER_MIMO_NOT_SUPPORT / PMI_WRONG / LAYER_WRONG


def oracle_tb_common(o, fmt, rnti, mcs, is2):
    """through o_ra_dl_dci_to_grant on a valid two-PRB localised allocation of a 50-PRB cell"""
    from lsn_testlib import OGrant
    cell = OCell(50, 2, 1, 1, 0, 0)
    d, g = ODciDl(), OGrant()
    d.rnti, d.format, d.alloc_type, d.riv, d.t2_nprb1a_is2 = rnti, fmt, 2, 50 * (2 - 1) + 4, is2
    d.tb[0].mcs_idx, d.tb[1].mcs_idx, d.tb[1].rv = mcs, 0, 1
    o.o_ra_dl_dci_to_grant.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    rc = o.o_ra_dl_dci_to_grant(C.byref(cell), 1, 2, 0, C.byref(d), C.byref(g))
    return (0 if rc == 0 else -1,) + ((g.nof_tb, g.tb[0].enabled, g.tb[1].enabled, g.tb[0].tbs, g.tb[0].mod) if rc == 0 else ())


def normalise_ul(row, mcs):
    """I_MCS 29..31 take modulation and size from the previous transmission (ul_sniffer_pusch.c:129-134); neither side has one here: the reference's empty
    last_tb reads as its enum value 0 (1 bit per symbol), the oracle writes 0, and for the CQI-only grant (I_MCS 29, CQI request, at most 4 PRB) the reference says
    QPSK with a transport block of 0 bits where the oracle leaves 0 - nothing is decoded in any of these cases (tbs 0).  Everything else is compared as it is."""
    if row is None or mcs < 29:
        return row
    return row[:4] + (0,) + row[5:]


SUITE_STRIDE = 23  # the suite walks every 23rd case of ul_sweep (110 k of 2.5 M); the generator walks all of them


def digest(rows):
    h = hashlib.sha256()
    for r in rows:
        h.update(repr(r).encode())
    return h.hexdigest()[:32]


def reference_sources_sha256(ref="/root/reference"):
    h = hashlib.sha256()
    for f in REF_SOURCES:
        h.update(open(os.path.join(ref, f), "rb").read())
    return h.hexdigest()
