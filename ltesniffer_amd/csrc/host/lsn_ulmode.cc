// lsn_ulmode.cc - UL_MODE commit stage: what SubframeWorker::run_ul_mode does after the DCI search
// (/root/reference/src/src/SubframeWorker.cc:236-347):
//   PDSCH_Decoder::decode_ul_mode      /root/reference/src/src/DL_Sniffer_PDSCH.cc:362-457   (RAR + format 1 / 1A, 64QAM table)
//   unpack_rar_response_ul_mode        /root/reference/src/src/DL_Sniffer_PDSCH.cc:632-671   (Msg3 grant of the RAR)
//   ULSchedule push / get              /root/reference/src/src/ULSchedule.cc:11-138          (DCI 0 -> PUSCH 4 subframes later, RAR grant 6)
//   PUSCH_Decoder::decode / decode_run /root/reference/src/src/UL_Sniffer_PUSCH.cc:250-310,389-583 (modulation-table trials, UL MCS tracking)
//   PDSCH_Decoder::decode_SIB          /root/reference/src/src/DL_Sniffer_PDSCH.cc:459-560   (until the uplink configuration is known)
// The uplink configuration comes from lsn_phy_set_ul_config / lsn_phy_set_prach_config or, like in the reference, from the first SIB2 seen.
// The whole function runs in the chunk's commit turn, i.e. strictly in TTI order; PUSCH attempts of a chunk are decoded in
// up to three batched waves (first / second / third attempt of the reference's trial order) and looked up by the sequential
// logic, which falls back to an on-demand decode when the MCS-tracking state moved inside the chunk.
#include "lsn_engine.h"
#include <map>
#include <tuple>

namespace lsn {

namespace {
// one srsran_pusch_decode call: the grant (sf, idx), the MCS table / modulation tried and the control-information layout it was tried with
// (uci: CQI report size | 1 + betaOffset indices << 8 / 13 / 18 - they move the UL-SCH resource elements, UL_Sniffer_PUSCH.cc:429-450)
struct Attempt { uint32_t sf; uint32_t idx; bool use256; int qm; uint32_t uci; };
inline bool operator<(const Attempt& a, const Attempt& b) { return std::tie(a.sf, a.idx, a.use256, a.qm, a.uci) < std::tie(b.sf, b.idx, b.use256, b.qm, b.uci); }
struct AttemptResult { bool crc = false; bool ran = false; float snr = 0.0f; std::vector<uint8_t> payload; };  // ran: the channel estimate of the attempt was computed
// downlink records keep the OFFSET of their payload in ch.h_payload: a later on-demand decode of the same loop may grow (reallocate) that
// arena, so the pointer is only formed when the record is emitted
struct PendingPdu { bool ul; char name; uint16_t rnti; uint8_t tb; size_t off; std::vector<uint8_t> own; uint32_t len; bool msg3 = false; };
}  // namespace

// PDSCH_Decoder::decode_SIB (DL_Sniffer_PDSCH.cc:459-560): every SI-RNTI grant of the subframe is decoded with the 64QAM table until a
// transport block carries a SystemInformation with SIB2; that block is the only record the subframe writes
bool Engine::decodeSib(Chunk& ch, JobRunner& r, uint32_t sf, Sib2Config& out, size_t& payload_off, uint32_t& len, uint8_t& tb_out)
{
  SubframeCtx& c = ch.ctx[sf];
  for (auto& e : c.dl) {
    if (e.rnti != SIRNTI || !e.unpack_ok) continue;
    if (e.job[0] < 0) e.job[0] = newJob(ch, sf, e, 0);
    const int j = e.job[0];
    if (j < 0) continue;
    if (!ch.jobs[j].done) { ensureJob(ch, r, j); r.perf.nof_ondemand_decodes++; }
    const DecodeJob& job = ch.jobs[j];
    for (int tb = 0; tb < 2; tb++) {
      if (!job.crc[tb]) continue;
      const uint32_t n = (uint32_t)(job.grant.tb[tb].tbs / 8);
      if (sib2_decode(ch.h_payload.data() + job.payload_off[tb], (int)n, out) == 2) { payload_off = (size_t)job.payload_off[tb]; len = n; tb_out = (uint8_t)tb; return true; }
    }
  }
  return false;
}

// MCSTracking::add_RNTI_ul (MCSTracking.cc:57-69): a new entry starts with unknown modulation and a copy of the default configuration
void Engine::ulTrackAdd(uint16_t rnti, int mod)
{
  if (ulmod[rnti]) return;  // std::map::insert keeps an existing entry
  ulmod[rnti] = (uint8_t)mod; ulmod_count++;
  ul_time[rnti] = commit_sf_cnt; ul_active[rnti] = ul_success[rnti] = 0;
  ul_uecfg[rnti] = mcs_tracking.default_config();
}

// MCSTracking::update_database_ul as LTESniffer_Core drives it (LTESniffer_Core.cc:473-499): entries idle for more than `interval` whole
// seconds (1 subframe = 1 ms) or without a counted decode are dropped
void Engine::ulAgeDatabase()
{
  if (ulmod.empty()) return;
  const uint32_t interval = mcs_update_period / 1000u;
  for (uint32_t r = 0; r < 65536; r++) {
    if (!ulmod[r]) continue;
    if ((commit_sf_cnt - ul_time[r]) / 1000u > interval || ul_active[r] == 0) { ulmod[r] = 0; ulmod_count--; }
  }
}

void Engine::commitChunkUl(Chunk& ch, JobRunner& r)
{
  // (the tracking vectors are sized in setCell; everything below that touches ulmod / ul_uecfg / ul_time / ul_active / ul_success / ulmod_count or
  // mcs_tracking holds mcs_mtx, because the caller's timer - update_database_ul, nof_RNTI_member_ul, get_ue_config_rnti through the C ABI - runs
  // on the caller's thread while chunks are in flight, LTESniffer_Core.cc:473-499)
  const uint32_t nsf = ch.nsf;
  syncUlConfig();
  std::vector<std::vector<PendingPdu>> out(nsf);            // records per subframe: downlink first, then uplink
  std::vector<std::vector<UlSchedGrant>> lists(nsf);        // PUSCH grants to try in each subframe
  std::vector<std::vector<std::pair<uint16_t, UeSpecConfig>>> setups(nsf);  // RRCConnectionSetups decoded in each subframe (run_decode, DL_Sniffer_PDSCH.cc:279-306)
  auto reconvert = [&](uint32_t from) {  // DCI 0 -> grant conversions done before the (re)configuration used another pusch-HoppingOffset
    for (uint32_t s = from; s < nsf; s++)
      for (auto& u : ch.ctx[s].ul) { u.finished = false; search->finishUlEntry(u); }
  };
  if (ch.ul_epoch != ul_cfg_epoch.load(std::memory_order_acquire)) reconvert(0);
  uint32_t prach_from = ul_set ? 0 : nsf;  // first subframe of this chunk that runs with a configuration

  // ---- phase 1 (sequential): downlink part + schedule bookkeeping ----
  for (uint32_t sf = 0; sf < nsf; sf++) {
    SubframeCtx& c = ch.ctx[sf];
    const uint32_t tti = c.tti % 10240;
    if (!ul_set) {  // SubframeWorker::run_ul_mode without a configuration (SubframeWorker.cc:238-252): decode_SIB and nothing else
      Sib2Config s2; size_t off = 0; uint32_t len = 0; uint8_t tb = 0;
      if (c.searched && decodeSib(ch, r, sf, s2, off, len, tb)) {
        out[sf].push_back({false, 'S', SIRNTI, tb, off, {}, len});
        // ULSchedule::set_config (ULSchedule.cc:140-158) + SubframeWorker.cc:256-282: DMRS, hopping offset, PRACH detector
        lsn_ul_cfg_t u{}; u.cyclic_shift = s2.cyclic_shift; u.delta_ss = s2.group_assignment_pusch; u.hopping_offset = s2.pusch_hop_offset;
        u.group_hopping_enabled = s2.group_hopping_enabled; u.sequence_hopping_enabled = s2.sequence_hopping_enabled;  // ULSchedule.cc:143-146
        if (setUlConfig(u) != LSN_SUCCESS) throw std::runtime_error("UL_MODE: the uplink configuration of SIB2 could not be applied");
        sib2 = s2; sib2_learned = true;
        lsn_prach_cfg_t pc{}; pc.config_idx = s2.prach_config_idx; pc.root_seq_idx = s2.root_seq_idx; pc.zero_corr_zone = s2.zero_corr_zone;
        pc.freq_offset = s2.prach_freq_offset; pc.hs_flag = s2.high_speed_flag;
        if (setPrachConfig(pc) != LSN_SUCCESS)
          fprintf(stderr, "ltesniffer_amd: the PRACH configuration of SIB2 (index %u, high-speed %u) is outside the supported set - no PRACH detection\n", s2.prach_config_idx, s2.high_speed_flag);
        reconvert(sf + 1);
        prach_from = sf + 1;
      }
      continue;
    }
    std::vector<UlSchedGrant> rar_now;
    if (c.searched) {
      for (auto& e : c.dl) {
        if (!e.unpack_ok || !ulModeDecodesDl(e)) continue;
        if (e.job[0] < 0) e.job[0] = newJob(ch, sf, e, 0);
        const int j = e.job[0];
        if (j < 0) continue;
        if (!ch.jobs[j].done) { ensureJob(ch, r, j); r.perf.nof_ondemand_decodes++; }
        const DecodeJob& job = ch.jobs[j];
        const bool is_ra = e.rnti >= RARNTI_START && e.rnti <= RARNTI_END;
        for (int tb = 0; tb < 2; tb++) {
          if (!job.crc[tb]) continue;
          const uint32_t len = (uint32_t)(job.grant.tb[tb].tbs / 8);
          out[sf].push_back({false, is_ra ? 'R' : rnti_name(e.rnti)[0], e.rnti, (uint8_t)tb, (size_t)job.payload_off[tb], {}, len});
          if (!is_ra) {  // betaOffset indices + aperiodic CQI mode of the UE, used by the PUSCH decoder from this subframe on
            UeSpecConfig sc[8];
            const int ns = MCSTracking::setups_of_pdu(ch.h_payload.data() + job.payload_off[tb], (int)len, sc, 8);
            for (int k = 0; k < ns; k++) setups[sf].push_back({e.rnti, sc[k]});
          }
          if (is_ra) {  // unpack_rar_response_ul_mode on TB 0's buffer; the grant of the LAST sub-header survives; then return
            RarEntry re[32];
            const int nre = rar_parse(cell, ch.h_payload.data() + job.payload_off[0], (int)len, re, 32);
            UlSchedGrant last; const bool found = nre > 0;
            if (found) {
              const RarEntry& e = re[nre - 1];
              last.rnti = e.t_crnti; last.is_rar = true; last.hopping = false;  // a hopping RAR grant is a type-1 grant (rar_parse)
              if (e.grant_ok) last.g = e.grant;  // ran_ul_grant_256 stays empty for RAR grants
              // (the RNTI activation of every sub-header already happened at search time)
            }
            if (found) rar_now.push_back(last);
            break;
          }
        }
      }
      // ULSchedule::pushULSche(tti, dci_ul): every accepted DCI 0, also those whose grant conversion failed
      std::vector<UlSchedGrant>& cur = ul_sched[tti];
      for (auto& u : c.ul) {
        UlSchedGrant g;
        g.rnti = u.rnti; g.n_dmrs = u.dci.n_dmrs; g.hopping = false;  // DCI 0 hopping lives in the grants (hop = 1 decoded, 2 not)
        g.cqi_req = u.dci.cqi_req != 0;
        for (auto& e : c.dl)  // "check nof_ack for uplink pusch decoder", SubframeWorker.cc:318-336
          if (e.rnti == u.rnti) {
            if (e.grant64.nof_tb == 1) g.nof_ack = 1;
            else if (e.grant64.nof_tb == 2) g.nof_ack = 2;
          }
        if (u.ok) { g.g = u.grant; g.g256 = u.grant256; }
        cur.push_back(g);
      }
    } else {
      ul_sched[tti];
    }
    rar_sched[tti] = rar_now;
    // PUSCH_Decoder::decode: grants of tti - 4 followed by the RAR grants of tti - 6
    const uint32_t t4 = (tti + 10240 - 4) % 10240, t6 = (tti + 10240 - 6) % 10240;
    auto ia = ul_sched.find(t4);
    if (ia != ul_sched.end()) { lists[sf] = ia->second; ul_sched.erase(ia); }
    auto ir = rar_sched.find(t6);
    if (ir != rar_sched.end()) { lists[sf].insert(lists[sf].end(), ir->second.begin(), ir->second.end()); rar_sched.erase(ir); }
  }
  // ---- PRACH occasions of the configured part of this chunk (PUSCH_Decoder::work_prach, UL_Sniffer_PUSCH.cc:656-713; no effect on the pcap) ----
  if (prach.set && ch.d_iq_src && prach_from < nsf) {
    std::vector<lsn_prach_det_t> det;
    prachDetectDev(ch.d_iq_src, cd.iq_nant, 1, nsf, ch.start_tti, det, prach_from);
    for (size_t a = 0; a < det.size();) {
      size_t b = a;
      while (b < det.size() && det[b].sf == det[a].sf) b++;
      if (prach_sink) prach_sink(prach_sink_user, (ch.start_tti + det[a].sf) % 10240u, det.data() + a, (uint32_t)(b - a));
      a = b;
    }
  }
  // stale schedule entries (a gap in the TTI sequence) must not accumulate
  while (ul_sched.size() > 64) ul_sched.erase(ul_sched.begin());
  while (rar_sched.size() > 64) rar_sched.erase(rar_sched.begin());

  // ---- the reference's trial order as a list of attempts, given the tracked maximum modulation ----
  auto valid_grant = [&](const UlSchedGrant& m) { return ulGrantValid(m.rnti, m.is_rar, m.g.tbs, m.g256.tbs, m.g.L_prb); };  // investigate_valid_ul_grant (lsn_lte.cc)
  auto mod_of = [&](uint16_t rnti) -> int { return ulmod[rnti] ? ulmod[rnti] : (ulmod_count < 250 ? 1 : 5); };  // find_tracking_info_RNTI_ul; 5 = FULL_BUFFER
  auto trial = [&](const UlSchedGrant& m, int mod, Attempt* a, int* learn) -> int {
    // returns the number of attempts (tried in order until one passes); learn[i]: modulation learnt if attempt i passes (0: none) - ulTrialPlan, lsn_lte.cc
    UlTry t[3];
    const int n = ulTrialPlan(m.g.mcs_idx, m.g.mod, m.g256.L_prb, m.g256.mod, mod, t);
    for (int i = 0; i < n; i++) { a[i].use256 = t[i].use256; a[i].qm = t[i].qm; learn[i] = t[i].learn; }
    return n;
  };
  // uci_cfg of an attempt, UL_Sniffer_PUSCH.cc:429-450: HARQ-ACK bits come with the grant; an aperiodic CSI request adds a CQI report of the
  // UE's configured type (srsran_cqi_size: wideband 4, UE-selected sub-band 4 + 1, higher-layer sub-band 4 + 2 N bits) and one RI bit
  auto uci_of = [&](const UlSchedGrant& m) -> uint32_t {
    const UeSpecConfig uc = ulUeConfig(m.rnti);
    uint32_t cqi = 0;
    if (m.cqi_req) {
      const uint32_t k = cell.nof_prb <= 7 ? 0u : (cell.nof_prb <= 26 ? 4u : (cell.nof_prb <= 63 ? 6u : 8u));  // dl_sniffer_pdsch.c:276-305
      cqi = uc.cqi_type == 0 ? 4u : (uc.cqi_type == 1 ? 5u : (k ? 4u + 2u * ((cell.nof_prb + k - 1) / k) : 0u));
    }
    return cqi | ((uc.i_offset_ack + 1u) & 31u) << 8 | ((uc.i_offset_cqi + 1u) & 31u) << 13 | ((uc.i_offset_ri + 1u) & 31u) << 18;
  };
  std::map<Attempt, AttemptResult> results;
  auto run_batch = [&](const std::vector<Attempt>& batch) {
    std::vector<lsn_pusch_grant_t> gl;
    std::vector<Attempt> keys;
    for (const Attempt& a : batch) {
      if (results.count(a)) continue;
      const UlSchedGrant& m = lists[a.sf][a.idx];
      const PuschGrant& g = a.use256 ? m.g256 : m.g;
      results[a];  // default: failed
      if (g.L_prb <= 2) r.perf.nof_pusch_on_unverified_dmrs++;  // tabulated reference signals (36.211 Tables 5.5.1.2-1 / -2): restated, structure-checked only
      if (m.hopping || g.hop == 2 || g.tbs <= 0) continue;  // type-2 hopping is not applied by the reference either (hopping_enabled stays false, SubframeWorker.cc:269)
      lsn_pusch_grant_t q{};
      q.sf = a.sf; q.rnti = m.rnti; q.n_dmrs = (uint16_t)m.n_dmrs; q.n_prb = g.n_prb; q.L_prb = g.L_prb; q.mod = (uint32_t)a.qm; q.tbs = (uint32_t)g.tbs; q.rv = g.rv;
      q.hop = g.hop; q.n_prb_slot1 = g.n_prb2;
      q.nof_ack = m.nof_ack;
      q.cqi_bits = a.uci & 255u; q.ri_bits = m.cqi_req ? 1u : 0u;
      q.beta_offset_ack_idx_p1 = (a.uci >> 8) & 31u; q.beta_offset_cqi_idx_p1 = (a.uci >> 13) & 31u; q.beta_offset_ri_idx_p1 = (a.uci >> 18) & 31u;
      gl.push_back(q);
      keys.push_back(a);
    }
    if (gl.empty()) return;
    std::vector<lsn_pusch_result_t> res(gl.size());
    std::vector<uint8_t> pay;
    puschDecodeGrid(ch.d_ul_grid, ch.nsf, ch.start_tti, gl.data(), (uint32_t)gl.size(), res.data(), pay);
    for (size_t i = 0; i < gl.size(); i++) {
      AttemptResult& ar = results[keys[i]];
      ar.crc = res[i].crc_ok != 0;
      ar.ran = res[i].iterations > 0; ar.snr = res[i].snr_db;
      if (ar.crc) ar.payload.assign(pay.begin() + res[i].payload_off, pay.begin() + res[i].payload_off + gl[i].tbs / 8);
      r.perf.nof_tb_decodes++;
      r.perf.nof_turbo_iterations += res[i].iterations;
    }
  };
  // ---- phase 2: three waves, predicted with the tracking state as of now ----
  for (int wave = 0; wave < 3; wave++) {
    std::vector<Attempt> batch;
    std::unique_lock<std::mutex> mcs_lk(mcs_mtx);
    for (uint32_t sf = 0; sf < nsf; sf++)
      for (uint32_t i = 0; i < lists[sf].size(); i++) {
        const UlSchedGrant& m = lists[sf][i];
        if (!valid_grant(m)) continue;
        Attempt a[3]; int learn[3];
        const int n = trial(m, mod_of(m.rnti), a, learn);
        if (wave >= n) continue;
        const uint32_t uci = uci_of(m);
        bool earlier_passed = false;
        for (int k = 0; k < wave; k++) { a[k].sf = sf; a[k].idx = i; a[k].uci = uci; auto it = results.find(a[k]); earlier_passed = earlier_passed || (it != results.end() && it->second.crc); }
        if (earlier_passed) continue;
        a[wave].sf = sf; a[wave].idx = i; a[wave].uci = uci;
        batch.push_back(a[wave]);
      }
    mcs_lk.unlock();
    run_batch(batch);
  }
  // ---- phase 3 (sequential): the exact decision logic, records in (tti, downlink, uplink) order ----
  for (uint32_t sf = 0; sf < nsf; sf++) {
    const uint32_t tti = ch.ctx[sf].tti;  // already reduced mod 10240 (SubframeCtx::reset)
    std::lock_guard<std::mutex> mcs_lk(mcs_mtx);  // per subframe, as commitChunk does
    if (cfg.mcs_tracking_mode && mcs_update_period && commit_sf_cnt && (commit_sf_cnt % mcs_update_period) == 0) ulAgeDatabase();  // LTESniffer_Core.cc:473-499
    for (auto& su : setups[sf]) {  // the downlink part of the subframe ran first (SubframeWorker.cc:299-347): update_default_ue_config / update_ue_config_rnti
      if (!mcs_tracking.check_default_config()) mcs_tracking.update_default_ue_config(su.second);
      ulTrackAdd(su.first);
      ul_uecfg[su.first] = su.second;
    }
    for (uint32_t i = 0; i < lists[sf].size(); i++) {
      const UlSchedGrant& m = lists[sf][i];
      if (!valid_grant(m)) continue;
      Attempt a[3]; int learn[3];
      const int mod = mod_of(m.rnti);
      if (ulmod[m.rnti]) ul_time[m.rnti] = commit_sf_cnt;  // find_tracking_info_RNTI_ul refreshes the entry's time stamp (MCSTracking.cc:52-53)
      const int n = trial(m, mod, a, learn);
      const uint32_t uci = uci_of(m), mcs = m.g.mcs_idx;
      const int mem_mod = (mcs > 20 && mcs < 29 && mod >= 2 && mod <= 4) ? mod : 1;  // decoding_mem.mcs_mod, UL_Sniffer_PUSCH.cc:456-570
      bool crc = false;
      for (int k = 0; k < n; k++) {
        a[k].sf = sf; a[k].idx = i; a[k].uci = uci;
        if (!results.count(a[k])) { run_batch({a[k]}); r.perf.nof_ondemand_decodes++; }
        const AttemptResult& ar = results[a[k]];
        if (ar.ran) last_ul_snr = ar.snr;
        if (!ar.crc) continue;
        crc = true;
        PendingPdu p{true, 'C', m.rnti, 0, 0, ar.payload, (uint32_t)ar.payload.size(), m.is_rar};
        out[sf].push_back(std::move(p));
        if (learn[k]) {  // decode_run: update_RNTI_ul (above MCS 20 when the maximum modulation was still unknown; below, after a 256QAM-table success)
          if (ulmod[m.rnti]) ulmod[m.rnti] = (uint8_t)learn[k];
          else ulTrackAdd(m.rnti);
        }
        break;
      }
      if (last_ul_snr >= 1.0f) {  // update_statistic_ul (MCSTracking.cc:729-754) behind the SNR gate of UL_Sniffer_PUSCH.cc:571-575
        ulTrackAdd(m.rnti, mem_mod);
        ul_active[m.rnti]++;
        if (crc) ul_success[m.rnti]++;
      }
    }
    for (auto& p : out[sf]) {
      if (p.ul) {  // write_ul_crnti, PcapWriter.cc:172-175: the payload joins the chunk's arena so that the writer thread finds it there
        r.perf.nof_pdus++;
        if (sink || api_mode >= 0) {
          lsn_pdu_ctx_t c{}; c.tti = tti; c.rnti = p.rnti; c.direction = 0; c.rnti_type = 3; c.crc_ok = 1;
          const size_t off = ch.h_payload.size();
          ch.h_payload.insert(ch.h_payload.end(), p.own.begin(), p.own.end());
          ch.recs.push_back({c, off, p.len, (uint8_t)(p.msg3 ? 1 : 0)});
        }
      } else {
        const char name[2] = {p.name, 0};
        emitPdu(ch, r, name, p.off, p.len, p.rnti, tti, p.tb);
      }
    }
    commit_sf_cnt++;
  }
}

}  // namespace lsn
