/* see srsran/standin.h: the two ASN.1 message types the reference's ULSchedule.h holds by value, down to the members DCICollection.h's closure names */
#pragma once
#include "srsran/standin.h"
#ifdef __cplusplus
namespace asn1 { namespace rrc {
struct pusch_cfg_basic_standin { uint32_t pusch_hop_offset = 0, n_sb = 0; };
struct ul_ref_sigs_pusch_standin { uint32_t cyclic_shift = 0, group_assign_pusch = 0; bool group_hop_enabled = false, seq_hop_enabled = false; };
struct pusch_cfg_common_standin { pusch_cfg_basic_standin pusch_cfg_basic; ul_ref_sigs_pusch_standin ul_ref_sigs_pusch; };
struct prach_cfg_info_standin { uint32_t prach_cfg_idx = 0, zero_correlation_zone_cfg = 0, prach_freq_offset = 0; bool high_speed_flag = false; };
struct prach_cfg_sib_standin { uint32_t root_seq_idx = 0; prach_cfg_info_standin prach_cfg_info; };
struct rr_cfg_common_standin { pusch_cfg_common_standin pusch_cfg_common; prach_cfg_sib_standin prach_cfg; };
struct sib_type2_s { rr_cfg_common_standin rr_cfg_common; };
/* RRCConnectionSetup-r8-IEs down to what PDSCH_Decoder::decode_rrc_connection_setup reads (DL_Sniffer_PDSCH.cc:129-179) */
struct cqi_report_mode_aperiodic_opts { enum options { rm12, rm20, rm22, rm30, rm31, nulltype } value; };
typedef cqi_report_mode_aperiodic_opts::options cqi_report_mode_aperiodic_e;
struct pusch_cfg_ded_standin { uint32_t beta_offset_ack_idx = 0, beta_offset_cqi_idx = 0, beta_offset_ri_idx = 0; };
struct pdsch_cfg_ded_standin { uint32_t p_a = 0; };
struct cqi_report_cfg_standin { bool cqi_report_mode_aperiodic_present = false; cqi_report_mode_aperiodic_e cqi_report_mode_aperiodic = cqi_report_mode_aperiodic_opts::rm12; };
struct phys_cfg_ded_standin { pusch_cfg_ded_standin pusch_cfg_ded; pdsch_cfg_ded_standin pdsch_cfg_ded; cqi_report_cfg_standin cqi_report_cfg; };
struct rr_cfg_ded_standin { phys_cfg_ded_standin phys_cfg_ded; };
struct rrc_conn_setup_r8_ies_s { rr_cfg_ded_standin rr_cfg_ded; };
} }
#endif
