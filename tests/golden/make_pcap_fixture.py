#!/usr/bin/env python3
"""Generates tests/golden/pcap_records.json from the reference's example captures
(/root/reference/pcap_file_example/*.pcap - the only golden data the reference ships for this path: they pin the
MAC-LTE (DLT 147) record framing that LTESniffer_pcap_writer::pack_and_write emits, PcapWriter.cc:93-118).
Run in the build container (the GPU box has no /root/reference); the JSON is committed."""
import json, struct, sys, os
src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/pcap_file_example"
out = {}
for name in ("ltesniffer_dl_mode.pcap", "ltesniffer_ul_mode.pcap", "api_collector.pcap"):
    data = open(os.path.join(src, name), "rb").read()
    recs, off = [], 24
    while off < len(data):
        ts, tu, il, ol = struct.unpack("<IIII", data[off:off + 16]); off += 16
        recs.append(data[off:off + il]); off += il
    # keep every distinct (direction, rnti type) combination + the first records, full bytes for short ones
    keep, seen = [], set()
    for i, r in enumerate(recs):
        key = (r[1], r[2])
        if i < 12 or key not in seen or len(r) < 40 and len(keep) < 60:
            seen.add(key); keep.append(r)
    out[name] = {"global_header": data[:24].hex(), "nof_records": len(recs),
                 "records": [r[:19 + 48].hex() for r in keep], "record_lens": [len(r) for r in keep]}
    # procedure-level golden data (real captures of the reference): every random-access response with the records that follow it, and the
    # PDU length of every record (= transport block sizes the reference's grant conversion produced)
    full = [dict(direction=r[1], rnti_type=r[2], rnti=(r[4] << 8) | r[5], tti=(((r[10] << 8) | r[11]) >> 4) * 10 + (((r[10] << 8) | r[11]) & 15), pdu=r[19:]) for r in recs]
    proc = []
    for i, r in enumerate(full):
        if r["rnti_type"] == 2:
            proc.append({"rar_tti": r["tti"], "ra_rnti": r["rnti"], "rar_pdu": r["pdu"].hex(),
                         "following": [dict(direction=q["direction"], rnti_type=q["rnti_type"], rnti=q["rnti"], dtti=(q["tti"] - r["tti"]) % 10240, pdu=q["pdu"].hex())
                                       for q in full[i + 1:i + 4]]})
    out[name]["random_access"] = proc
    out[name]["pdu_lengths"] = {"%d/%d" % (d, t): sorted({len(q["pdu"]) for q in full if q["direction"] == d and q["rnti_type"] == t})
                                for d, t in sorted({(q["direction"], q["rnti_type"]) for q in full})}
    # downlink C-RNTI MAC PDUs: the contention-resolution messages (a CCCH SDU = RRCConnectionSetup behind a contention resolution
    # identity) in full, and (lcid, length) lists of a sample of the others, walked by the independent parser below
    def walk(p):
        pos, subs, more = 0, [], True
        while more:
            b = p[pos]; pos += 1
            lcid, more, ln = b & 31, bool(b & 32), None
            if lcid < 26 and more:
                ln = p[pos] & 127
                if p[pos] & 128:
                    ln = (ln << 8) | p[pos + 1]; pos += 1
                pos += 1
            subs.append([lcid, ln])
        for s_ in subs:
            if s_[0] >= 26:
                s_[1] = {28: 6, 29: 1, 27: 1}.get(s_[0], 0)
        if subs[-1][1] is None:
            subs[-1][1] = len(p) - pos - sum(x[1] for x in subs[:-1])
        return subs
    dl = [q for q in full if q["direction"] == 1 and q["rnti_type"] == 3]
    out[name]["dl_crnti_walks"] = [dict(pdu_head=q["pdu"][:24].hex(), length=len(q["pdu"]), subheaders=walk(q["pdu"])) for q in dl[:60]]
    # BCCH-DL-SCH messages (SI-RNTI records): in UL_MODE the reference writes exactly the SystemInformation it took SIB2 from (decode_SIB)
    out[name]["si_pdus"] = sorted({q["pdu"].hex() for q in full if q["rnti_type"] == 4})
    # uplink Msg3 blocks (7 bytes: CCCH sub-header + RRCConnectionRequest) the reference's API collected next to the connection setups
    out[name]["msg3"] = [dict(rnti=q["rnti"], pdu=q["pdu"].hex()) for q in full if q["direction"] == 0 and len(q["pdu"]) == 7 and q["pdu"][0] == 0x00]
    # the uplink SRB blocks the reference's API parsers accepted (api_collector.pcap only: UE capability information, attach requests)
    if name == "api_collector.pcap":
        out[name]["ul_dcch"] = [dict(rnti=q["rnti"], pdu=q["pdu"].hex()) for q in full if q["direction"] == 0 and len(q["pdu"]) > 7]
    out[name]["conn_setup"] = [dict(rnti=q["rnti"], pdu=q["pdu"].hex()) for q in dl if len(q["pdu"]) > 8 and q["pdu"][0] == 0x3C and (q["pdu"][1] & 31) == 0]
    # downlink SRB1 blocks (first sub-header LCID 1 with a length field): the DL-DCCH messages of an attach - dlInformationTransfer, securityModeCommand,
    # ueCapabilityEnquiry, the RRCConnectionReconfiguration that carries the attach accept, rrcConnectionRelease
    out[name]["dl_dcch"] = [dict(rnti=q["rnti"], tti=q["tti"], pdu=q["pdu"].hex()) for q in dl if len(q["pdu"]) > 8 and q["pdu"][0] == 0x21 and q["pdu"][1] > 3]
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "pcap_records.json"), "w"), indent=0)
print({k: (v["nof_records"], len(v["records"])) for k, v in out.items()})
