#!/bin/bash
# session 21: how much of the chains' latency is host reaction time?  LSN_SPIN_WAIT=1 (hipEventSynchronize instead of query + 50 us naps in every pipeline wait)
cd ${GRAFT_REPO_ROOT:-.}
EXP_STEPS=3 EXP_WARMUP=2 bash tools/r5_exp.sh r05d_session21 'base (query + 50 us naps)||' 'spin|LSN_SPIN_WAIT=1|' 'base||' 'spin|LSN_SPIN_WAIT=1|' 'base 16 dB||--workload cfg3_at_16_dB_snr' 'spin 16 dB|LSN_SPIN_WAIT=1|--workload cfg3_at_16_dB_snr' | cut -c1-200
