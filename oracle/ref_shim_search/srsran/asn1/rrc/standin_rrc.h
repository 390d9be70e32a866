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
struct rrc_conn_setup_r8_ies_s { int _ = 0; };
} }
#endif
