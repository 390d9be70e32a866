/* see srsran/standin.h */
#pragma once
#include "srsran/standin.h"
#ifdef __cplusplus
extern "C" {
#endif
float srsran_vec_avg_power_cf(const cf_t* x, const uint32_t len); /* SubframePower.cc (ul_decode_glue.cc supplies the plain mean of |x|^2) */
#ifdef __cplusplus
}
#endif
