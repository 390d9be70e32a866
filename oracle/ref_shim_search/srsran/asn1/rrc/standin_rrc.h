/* see srsran/standin.h: the two ASN.1 message types the reference's ULSchedule.h holds by value, down to the members DCICollection.h's closure names */
#pragma once
#include "srsran/standin.h"
#ifdef __cplusplus
namespace asn1 { namespace rrc {
struct pusch_cfg_basic_standin { uint32_t pusch_hop_offset = 0, n_sb = 0; };
struct pusch_cfg_common_standin { pusch_cfg_basic_standin pusch_cfg_basic; };
struct rr_cfg_common_standin { pusch_cfg_common_standin pusch_cfg_common; };
struct sib_type2_s { rr_cfg_common_standin rr_cfg_common; };
struct rrc_conn_setup_r8_ies_s { int _ = 0; };
} }
#endif
