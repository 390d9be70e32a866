/* Declarations (no definitions) of the srsRAN radio / synchronisation / MIB API that /root/reference/src/src/LTESniffer_Core.cc names.  They exist for ONE check,
 * tests/test_reference_caller.py: that the reference's own caller translation unit compiles, syntax and types, against include/ltesniffer_amd_compat.hpp.
 * srsRAN is absent from /root/reference and from this image and these functions are outside the path (ue_sync, RF and cell search stay
 * the caller's); nothing is linked or run, nothing here is an implementation, and no parity claim rests on it.  Members: only those the caller touches. */
#pragma once
#include "srsran/standin.h"   /* oracle/ref_shim_search: cf_t, srsran_cell_t, srsran_dl_sf_cfg_t, srsran_pdsch_cfg_t, RNTI ranges ... */
#ifdef __cplusplus
extern "C" {
#endif
#define SRSRAN_SF_LEN_PRB(nof_prb) (15 * ((nof_prb) <= 6 ? 128 : (nof_prb) <= 15 ? 256 : (nof_prb) <= 25 ? 384 : (nof_prb) <= 50 ? 768 : (nof_prb) <= 75 ? 1024 : 1536))
#define SRSRAN_DEFAULT_MAX_FRAMES_PBCH 500
#define SRSRAN_DEFAULT_MAX_FRAMES_PSS 10
#define SRSRAN_DEFAULT_NOF_VALID_PSS_FRAMES 10
#define SRSRAN_BCH_PAYLOAD_LEN 24
#define SRSRAN_UE_MIB_FOUND 1
#define SRSRAN_MAX_CODEBOOKS 4
#define SRSRAN_AGC_CALLBACK(NAME) void(NAME)(void* h, float gain_db)
#define SRSRAN_VERBOSE_NONE 0
typedef struct { uint32_t max_frames_pbch, max_frames_pss, nof_valid_pss_frames; float init_agc; bool force_tdd; } cell_search_cfg_t;
typedef struct { int unused; } srsran_rf_t;
typedef struct { double min_rx_gain, max_rx_gain; } srsran_rf_info_t;
typedef struct { time_t full_secs; double frac_secs; } srsran_timestamp_t;
typedef struct { int unused; } srsran_sync_t;
typedef struct { int unused; } srsran_pbch_t;
typedef struct { srsran_pbch_t pbch; } srsran_ue_mib_t;
typedef struct {
  float cfo_current_value;
  bool cfo_is_copied, cfo_correct_enable_find, cfo_correct_enable_track;
  srsran_sync_t sfind, strack;
  uint32_t frame_total_cnt;
  int state;
} srsran_ue_sync_t;
typedef enum { SRSRAN_PHICH_NORM = 0, SRSRAN_PHICH_EXT } srsran_phich_length_standin_t;
typedef enum { SRSRAN_PHICH_R_1_6 = 0, SRSRAN_PHICH_R_1_2, SRSRAN_PHICH_R_1, SRSRAN_PHICH_R_2 } srsran_phich_r_standin_t;
typedef int srsran_chest_dl_estimator_alg_t;
typedef struct { bool cfo_estimate_enable; uint32_t cfo_estimate_sf_mask; srsran_chest_dl_estimator_alg_t estimator_alg; bool sync_error_enable; } srsran_chest_dl_cfg_t;
srsran_chest_dl_estimator_alg_t srsran_chest_dl_str2estimator_alg(const char* str);
typedef int (ue_sync_recv_callback_t)(void*, cf_t* [SRSRAN_MAX_PORTS], uint32_t, srsran_timestamp_t*);
void* srsran_vec_malloc(uint32_t size);
int srsran_sampling_freq_hz(uint32_t nof_prb);
void srsran_cell_fprint(FILE* stream, srsran_cell_t* cell, uint32_t sfn);
extern int srsran_verbose;
void set_srsran_verbose_level(int level);
int get_srsran_verbose_level(void);
/* radio */
int srsran_rf_open_multi(srsran_rf_t* h, char* args, uint32_t nof_channels);
int srsran_rf_close(srsran_rf_t* h);
int srsran_rf_start_gain_thread(srsran_rf_t* rf, bool tx_gain_same_rx);
int srsran_rf_set_rx_gain(srsran_rf_t* h, double gain);
int srsran_rf_set_rx_gain_th(srsran_rf_t* h, double gain);
double srsran_rf_get_rx_gain(srsran_rf_t* h);
srsran_rf_info_t* srsran_rf_get_info(srsran_rf_t* h);
double srsran_rf_set_rx_freq(srsran_rf_t* h, uint32_t ch, double freq);
double srsran_rf_set_rx_srate(srsran_rf_t* h, double freq);
int srsran_rf_start_rx_stream(srsran_rf_t* h, bool now);
int srsran_rf_stop_rx_stream(srsran_rf_t* h);
void srsran_rf_flush_buffer(srsran_rf_t* h);
int srsran_rf_recv_with_time_multi(srsran_rf_t* h, void* data[SRSRAN_MAX_PORTS], uint32_t nsamples, bool blocking, time_t* secs, double* frac_secs);
int rf_search_and_decode_mib(srsran_rf_t* rf, uint32_t nof_rx_channels, cell_search_cfg_t* config, int force_N_id_2, srsran_cell_t* cell, float* cfo);
/* synchronisation */
int srsran_ue_sync_init_file_multi(srsran_ue_sync_t* q, uint32_t nof_prb, char* file_name, int offset_time, float offset_freq, uint32_t nof_rx_ant);
int srsran_ue_sync_init_multi_decim(srsran_ue_sync_t* q, uint32_t max_prb, bool search_cell, ue_sync_recv_callback_t* recv_callback, uint32_t nof_rx_antennas, void* stream_handler, int decimate);
int srsran_ue_sync_set_cell(srsran_ue_sync_t* q, srsran_cell_t cell);
void srsran_ue_sync_free(srsran_ue_sync_t* q);
int srsran_ue_sync_start_agc(srsran_ue_sync_t* q, SRSRAN_AGC_CALLBACK(set_gain_callback), float min_gain, float max_gain, float init_gain_value);
int srsran_ue_sync_zerocopy(srsran_ue_sync_t* q, cf_t* input_buffer[SRSRAN_MAX_PORTS], const uint32_t max_num_samples);
uint32_t srsran_ue_sync_get_sfidx(srsran_ue_sync_t* q);
float srsran_ue_sync_get_cfo(srsran_ue_sync_t* q);
float srsran_ue_sync_get_sfo(srsran_ue_sync_t* q);
int srsran_ue_sync_get_last_sample_offset(srsran_ue_sync_t* q);
void srsran_sync_set_cfo_cp_enable(srsran_sync_t* q, bool enable, uint32_t nof_symbols);
float srsran_sync_get_peak_value(srsran_sync_t* q);
/* MIB */
int srsran_ue_mib_init(srsran_ue_mib_t* q, cf_t* in_buffer, uint32_t max_prb);
int srsran_ue_mib_set_cell(srsran_ue_mib_t* q, srsran_cell_t cell);
void srsran_ue_mib_free(srsran_ue_mib_t* q);
int srsran_ue_mib_decode(srsran_ue_mib_t* q, uint8_t bch_payload[SRSRAN_BCH_PAYLOAD_LEN], uint32_t* nof_tx_ports, int* sfn_offset);
void srsran_pbch_decode_reset(srsran_pbch_t* q);
void srsran_pbch_mib_unpack(uint8_t* msg, srsran_cell_t* cell, uint32_t* sfn);
#ifdef __cplusplus
}
#endif
