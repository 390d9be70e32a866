// stage_c.hip - gfx950 kernels for the data-plane half of the LTESniffer worker: PDSCH RE gather + equalisation
// (single port MRC, SFBC, large-delay CDD / closed-loop spatial multiplexing with 2x2 MMSE) + soft demodulation +
// descrambling, and the per-code-block turbo decoder (gather rate de-matching, windowed max-log-MAP with
// next-iteration boundary init, early stop on the code-block CRC).  They replace srsran_ue_dl_decode_pdsch as called
// from /root/reference/src/src/DL_Sniffer_PDSCH.cc:997,1110,1207 (config /root/reference/src/src/SubframeWorker.cc:362-371).
// Mapping: demod = one thread per resource element (coalesced float2 loads of grid / channel estimates, int16 LLR
// stores); turbo = one wavefront per code block, lane = trellis window, all soft data of the block staged in LDS,
// forward metrics check-pointed every 16 steps and recomputed so that the block fits 2 workgroups per CU.
#include "lsn_dev.h"
#include <algorithm>
#include "lsn_rm.h"
#include <type_traits>

#define SQRT1_2F 0.70710678118654752440f
#define SQRT2F 1.41421356237309504880f
#define LLR_Q 180.0f

__device__ __forceinline__ cf32 cmulconj(cf32 a, cf32 b) { cf32 c; c.r = a.r * b.r + a.i * b.i; c.i = a.i * b.r - a.r * b.i; return c; }
__device__ __forceinline__ float cabs2(cf32 a) { return a.r * a.r + a.i * a.i; }

// ------------------------------------------------------------------------------------------------ RE bookkeeping
// prefix[l][prb] = number of PDSCH REs of this grant in symbol l before PRB prb; prefix[14*nprb + l] = REs before symbol l.
// One wavefront per job: lanes are PRBs (two rounds cover 110), the exclusive prefix over the PRBs of a symbol is a shuffle scan.
__global__ __launch_bounds__(64) void k_pdsch_prep(LsnCellDev c, const LsnGrantDev* __restrict__ jobs, uint16_t* __restrict__ prefix)
{
  const LsnGrantDev& g = jobs[blockIdx.x];
  const int lane = threadIdx.x, nprb = (int)c.nof_prb;
  uint16_t* pf = prefix + g.prefix_off;
  const int cls = g.sf_idx == 0 ? 0 : (g.sf_idx == 5 ? 1 : 2);
  uint32_t before = 0;  // REs of the symbols in front of l (wave-uniform)
  for (int l = 0; l < 14; l++) {
    uint32_t run = 0;   // REs of this symbol in front of the current round
    for (int base = 0; base < nprb; base += 64) {
      const int prb = base + lane;
      uint32_t v = 0;
      if (prb < nprb && l >= (int)g.l0 && ((g.prb_mask[l >= (int)c.nslot ? 1 : 0][prb >> 5] >> (prb & 31)) & 1u)) v = (uint32_t)__popc((unsigned)c.validmask[(cls * 14 + l) * nprb + prb]);
      uint32_t inc = v;  // inclusive scan over the 64 lanes
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
      }
      if (prb < nprb) pf[l * nprb + prb] = (uint16_t)(run + inc - v);
      run += __shfl(inc, 63);
    }
    if (lane == 0) pf[14 * nprb + l] = (uint16_t)before;
    before += run;
  }
  if (lane == 0) pf[14 * nprb + 14] = (uint16_t)before;
}
void lsn_launch_pdsch_prep(const LsnCellDev& c, const LsnGrantDev* g, uint16_t* prefix, uint32_t njobs, hipStream_t s)
{
  LSN_LAUNCH(k_pdsch_prep, dim3(njobs), dim3(64), 0, s, c, g, prefix);
}

// The same with the upload of the launch's descriptors folded in (one launch less in every decode chain - the chains' depth, not their work, bounds
// the engine: DESIGN 3.2): workgroup j < njobs reads job j from the PINNED HOST array, stores the device copy the later kernels read and computes its
// prefix table; the workgroups behind copy the other descriptor arrays (work items, code blocks) host -> device.
__global__ __launch_bounds__(64) void k_pdsch_prep_up(LsnCellDev c, const LsnGrantDev* __restrict__ jobs_h, LsnGrantDev* __restrict__ jobs_d, uint32_t njobs,
                                                      LsnCopySegs sg, uint16_t* __restrict__ prefix)
{
  const int lane = threadIdx.x, nprb = (int)c.nof_prb;
  if (blockIdx.x >= njobs) {
    const uint32_t w = blockIdx.x - njobs, nw = gridDim.x - njobs;
    for (uint32_t q = 0; q < sg.n; q++) {
      const uint32_t words = sg.words[q];
      const bool al = ((((uintptr_t)sg.src[q]) | ((uintptr_t)sg.dst[q])) & 15u) == 0;
      const uint32_t n16 = al ? words / 4 : 0;
      const uint4* a4 = (const uint4*)sg.src[q];
      uint4* b4 = (uint4*)sg.dst[q];
      for (uint32_t i = w * 64u + (uint32_t)lane; i < n16; i += nw * 64u) b4[i] = a4[i];
      const uint32_t* a = (const uint32_t*)sg.src[q];
      uint32_t* b = (uint32_t*)sg.dst[q];
      for (uint32_t i = n16 * 4 + w * 64u + (uint32_t)lane; i < words; i += nw * 64u) b[i] = a[i];
    }
    return;
  }
  __shared__ LsnGrantDev gs;
  constexpr int NW = (int)(sizeof(LsnGrantDev) / 4);
  static_assert(NW <= 64 && sizeof(LsnGrantDev) % 4 == 0, "one word per lane");
  if (lane < NW) {
    const uint32_t v = ((const uint32_t*)(jobs_h + blockIdx.x))[lane];
    ((uint32_t*)&gs)[lane] = v;
    ((uint32_t*)(jobs_d + blockIdx.x))[lane] = v;
  }
  __syncthreads();
  const LsnGrantDev& g = gs;
  uint16_t* pf = prefix + g.prefix_off;
  const int cls = g.sf_idx == 0 ? 0 : (g.sf_idx == 5 ? 1 : 2);
  uint32_t before = 0;
  for (int l = 0; l < 14; l++) {
    uint32_t run = 0;
    for (int base = 0; base < nprb; base += 64) {
      const int prb = base + lane;
      uint32_t v = 0;
      if (prb < nprb && l >= (int)g.l0 && ((g.prb_mask[l >= (int)c.nslot ? 1 : 0][prb >> 5] >> (prb & 31)) & 1u)) v = (uint32_t)__popc((unsigned)c.validmask[(cls * 14 + l) * nprb + prb]);
      uint32_t inc = v;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
      }
      if (prb < nprb) pf[l * nprb + prb] = (uint16_t)(run + inc - v);
      run += __shfl(inc, 63);
    }
    if (lane == 0) pf[14 * nprb + l] = (uint16_t)before;
    before += run;
  }
  if (lane == 0) pf[14 * nprb + 14] = (uint16_t)before;
}
void lsn_launch_pdsch_prep_up(const LsnCellDev& c, const LsnGrantDev* jobs_host, LsnGrantDev* jobs_dev, uint32_t njobs, const LsnCopySegs& sg, uint16_t* prefix, hipStream_t s)
{
  uint32_t total = 0;
  for (uint32_t q = 0; q < sg.n; q++) total += sg.words[q];
  const uint32_t ncopy = total ? std::max<uint32_t>(1u, std::min<uint32_t>((total / 4 + 63u) / 64u, 256u)) : 0u;
  LSN_LAUNCH(k_pdsch_prep_up, dim3(njobs + ncopy), dim3(64), 0, s, c, jobs_host, jobs_dev, njobs, sg, prefix);
}

// ------------------------------------------------------------------------------------------------ soft demodulation
__device__ __forceinline__ void demod_llr(int Qm, float I, float Q, float* L)
{
  float aI = fabsf(I), aQ = fabsf(Q);
  L[0] = -I; L[1] = -Q;
  if (Qm == 4) {
    const float a = 0.31622776601683794f;
    L[2] = aI - 2.0f * a; L[3] = aQ - 2.0f * a;
  } else if (Qm == 6) {
    const float a = 0.15430334996209191f;
    float tI = aI - 4.0f * a, tQ = aQ - 4.0f * a;
    L[2] = tI; L[3] = tQ; L[4] = fabsf(tI) - 2.0f * a; L[5] = fabsf(tQ) - 2.0f * a;
  } else if (Qm == 8) {
    const float a = 0.07669649888473704f;
    float tI = aI - 8.0f * a, tQ = aQ - 8.0f * a;
    float uI = fabsf(tI) - 4.0f * a, uQ = fabsf(tQ) - 4.0f * a;
    L[2] = tI; L[3] = tQ; L[4] = uI; L[5] = uQ; L[6] = fabsf(uI) - 2.0f * a; L[7] = fabsf(uQ) - 2.0f * a;
  }
}

// QM is a compile-time constant so that the soft-bit array stays in registers (a run-time loop bound puts it in scratch memory).
// Round 6 (last session): the QM positions of a resource element start at an even n0 = idx QM, so the scrambling tables are read with wide loads (x1 bytes: 2 / 4 / 8 at
// once, x2 masks: 8 or 16 bytes per load), the soft bits leave in 4- / 8- / 16-byte stores, the scrambling bit flips the SIGN BIT of the scaled value in front of the
// rounding (rint and the symmetric clip commute with a sign change, so -clip(rint(v)) = clip(rint(-v)) bit for bit, NaN included), and the clip is one integer
// v_med3 behind the conversion (v_cvt_i32_f32 saturates and turns NaN into 0 - what the compare / select pairs in front of it gave): 10 instead of 22 vector
// instructions per soft bit.
__device__ __forceinline__ int lsn_cvt_sat(float r)
{
  int q;
  asm("v_cvt_i32_f32 %0, %1" : "=v"(q) : "v"(r));
  return q;
}
template <int QM>
__device__ __forceinline__ void emit_q(const LsnCellDev& c, cf32 x, float w, float inv_amp, uint32_t cinit, uint32_t idx, int16_t* __restrict__ out)
{
  float L[8];
  const float wq = w * LLR_Q;
  demod_llr(QM, x.r * inv_amp, x.i * inv_amp, L);
  const uint32_t n0 = idx * (uint32_t)QM;
  uint32_t x1w[2] = {0u, 0u}, m[8];
  const uint8_t* p1 = c.gold_x1 + n0;
  const uint32_t* p2 = c.gold_x2mask + n0;
  if (QM == 8) {
    const uint2 v = *(const uint2*)p1; x1w[0] = v.x; x1w[1] = v.y;
    const uint4 a = *(const uint4*)p2, b = *(const uint4*)(p2 + 4);
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w; m[4] = b.x; m[5] = b.y; m[6] = b.z; m[7] = b.w;
  } else if (QM == 6) {
    const uint16_t* h = (const uint16_t*)p1;
    x1w[0] = (uint32_t)h[0] | ((uint32_t)h[1] << 16); x1w[1] = (uint32_t)h[2];
    const uint2 a = *(const uint2*)p2, b = *(const uint2*)(p2 + 2), d = *(const uint2*)(p2 + 4);
    m[0] = a.x; m[1] = a.y; m[2] = b.x; m[3] = b.y; m[4] = d.x; m[5] = d.y;
  } else if (QM == 4) {
    x1w[0] = *(const uint32_t*)p1;
    const uint4 a = *(const uint4*)p2;
    m[0] = a.x; m[1] = a.y; m[2] = a.z; m[3] = a.w;
  } else {
    x1w[0] = (uint32_t)*(const uint16_t*)p1;
    const uint2 a = *(const uint2*)p2;
    m[0] = a.x; m[1] = a.y;
  }
  uint32_t pk[4];
#pragma unroll
  for (int b = 0; b < QM; b += 2) {
    int q[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int bb = b + h;
      const uint32_t sign = ((uint32_t)__popc(m[bb] & cinit) ^ (x1w[bb >> 2] >> (8 * (bb & 3)))) << 31;   // only bit 0 of the sum survives the shift
      const float v = __uint_as_float(__float_as_uint(L[bb] * wq) ^ sign);
      const int t = lsn_cvt_sat(rintf(v));
      q[h] = t < -LSN_LLR_CLIP ? -LSN_LLR_CLIP : (t > LSN_LLR_CLIP ? LSN_LLR_CLIP : t);
    }
    pk[b >> 1] = __builtin_amdgcn_perm((uint32_t)q[1], (uint32_t)q[0], 0x05040100u);
  }
  uint32_t* o = (uint32_t*)(out + n0);   // llr_off is a multiple of 8 entries, n0 is even (QM = 8: a multiple of 8, QM = 4: of 4)
  if (QM == 8) *(uint4*)o = make_uint4(pk[0], pk[1], pk[2], pk[3]);
  else if (QM == 6) { o[0] = pk[0]; o[1] = pk[1]; o[2] = pk[2]; }
  else if (QM == 4) *(uint2*)o = make_uint2(pk[0], pk[1]);
  else o[0] = pk[0];
}
__device__ __forceinline__ void emit(const LsnCellDev& c, int Qm, cf32 x, float w, float inv_amp, uint32_t cinit, uint32_t idx,
                                     int16_t* __restrict__ out)
{
  switch (Qm) {
    case 2: emit_q<2>(c, x, w, inv_amp, cinit, idx, out); break;
    case 4: emit_q<4>(c, x, w, inv_amp, cinit, idx, out); break;
    case 6: emit_q<6>(c, x, w, inv_amp, cinit, idx, out); break;
    case 8: emit_q<8>(c, x, w, inv_amp, cinit, idx, out); break;
    default: break;
  }
}

// grid = (work items, 14 symbols): an item is one group of 16 PRBs inside the PRB span of one job (items[i] = job << 8 | group), listed by the
// host - a grid over all groups of all jobs would be mostly empty workgroups (a grant covers a few PRBs of the 100)
__global__ __launch_bounds__(192) void k_pdsch_demod(LsnCellDev c, const LsnGrantDev* __restrict__ jobs, const uint32_t* __restrict__ items,
                                                     const uint16_t* __restrict__ prefix,
                                                     const cf32* __restrict__ grid, const cf32* __restrict__ ce,
                                                     const LsnChest* __restrict__ chest, int16_t* __restrict__ llr)
{
  const uint32_t item = items[blockIdx.x];
  const LsnGrantDev& g = jobs[item >> 8];
  const int nprb = (int)c.nof_prb, nre = (int)c.nre, A = (int)c.nof_rx;
  const int l = blockIdx.y, prb = (int)(item & 255u) * 16 + (int)threadIdx.x / 12, kk = (int)threadIdx.x % 12;
  if (l < (int)g.l0 || prb >= nprb) return;
  if (!((g.prb_mask[l >= (int)c.nslot ? 1 : 0][prb >> 5] >> (prb & 31)) & 1u)) return;
  const int cls = g.sf_idx == 0 ? 0 : (g.sf_idx == 5 ? 1 : 2);
  const unsigned mask = c.validmask[(cls * 14 + l) * nprb + prb];
  if (!((mask >> kk) & 1u)) return;
  const uint16_t* pf = prefix + g.prefix_off;
  const uint32_t idx = (uint32_t)pf[14 * nprb + l] + (uint32_t)pf[l * nprb + prb] + (uint32_t)__popc(mask & ((1u << kk) - 1u));
  const int k = 12 * prb + kk;
  const LsnChest ch = chest[g.sf];
  const float noise = ch.noise_avg, chan_ref = ch.chan_ref;
  const int lq = l >= (int)c.nslot ? l - (int)c.nslot : l;
  const float inv_amp = (lq == 0 || lq == (int)c.nslot - 3) ? g.inv_amp_b : g.inv_amp_a;   // rho_B on the symbols with the CRS of ports 0, 1 (36.213 Table 5.2-2)
  const cf32* gr = grid + (size_t)g.sf * A * 14 * nre;                 // (wave-uniform: scalar arithmetic)
  const cf32* ch0 = ce + (size_t)g.sf * c.nof_ports * A * 14 * nre;
  // Element offsets inside the subframe's planes in 32 bits with 24-bit multiplies (round 6, last session): the size_t products of rounds 1-5 compiled to
  // three multiply instructions (v_mad_u64_u32 + 2 v_mul_lo_u32) per address wherever the port index is a per-lane value (transmit diversity), and to
  // vector instructions on wave-uniform values elsewhere: 15.85 M -> 15.20 M vector instructions per launch.  The largest offset is 8 planes x 14 x 1 320 elements.
  const uint32_t plane = 14u * (uint32_t)nre, rowoff = (uint32_t)l * (uint32_t)nre;
#define GRID(rx, kq) gr[__umul24((uint32_t)(rx), plane) + rowoff + (uint32_t)(kq)]
#define CE(p, rx, kq) ch0[__umul24(__umul24((uint32_t)(p), (uint32_t)A) + (uint32_t)(rx), plane) + rowoff + (uint32_t)(kq)]
  int16_t* out0 = llr + g.llr_off[0];
  int16_t* out1 = llr + g.llr_off[1];
  switch (g.tx_scheme) {
    case 0: {  // single antenna port
      float nr = 0, ni = 0, den = 0;
      for (int rx = 0; rx < A; rx++) {
        cf32 h = CE(0, rx, k);
        cf32 t = cmulconj(GRID(rx, k), h);
        float hp = cabs2(h);
        if (rx == 0) { nr = t.r; ni = t.i; den = hp; } else { nr = nr + t.r; ni = ni + t.i; den = den + hp; }
      }
      float dn = den + noise;
      cf32 x; x.r = nr / dn; x.i = ni / dn;
      emit(c, (int)g.qm[0], x, den / chan_ref, inv_amp, g.cinit[0], idx, out0);
      break;
    }
    case 1: {  // transmit diversity (SFBC; SFBC-FSTD with four ports), pairs of consecutive REs of the mapping order
      // A pair never leaves its PRB here: an RE without partner inside the PRB (odd number of PDSCH REs in a PRB of this symbol) carries zero soft
      // bits, as in the oracle's zero-initialised buffer.  The kernel writes those zeros itself (rounds 1-3 cleared the whole arena in front of
      // every launch: 0.29 MB of HBM traffic per subframe and one fill kernel per decode launch).
      const unsigned hi = mask >> (kk + 1);
      const bool lone = (idx & 1u) ? (mask & ((1u << kk) - 1u)) == 0u : hi == 0u;
      if (lone) {
        const uint32_t n0 = idx * g.qm[0];
        for (uint32_t b = 0; b < g.qm[0]; b++) out0[n0 + b] = 0;
        return;
      }
      if (idx & 1u) return;
      const int k2 = k + 1 + (__ffs(hi) - 1);
      float x0r = 0, x0i = 0, x1r = 0, x1i = 0, hh = 0;
      // four ports (SFBC-FSTD, 36.211 6.3.4.3): symbol pairs alternate between the port pairs (0, 2) and (1, 3); each pair sees half of the ports
      // chan_ref sums over, so its weight doubles
      const bool fstd = c.nof_ports == 4;
      const int pa = (fstd && (idx & 2u)) ? 1 : 0, pb = fstd ? pa + 2 : 1;
      for (int rx = 0; rx < A; rx++) {
        cf32 r0 = GRID(rx, k), r1 = GRID(rx, k2);
        cf32 h00 = CE(pa, rx, k), h01 = CE(pa, rx, k2), h10 = CE(pb, rx, k), h11 = CE(pb, rx, k2);
        float hp = cabs2(h00) + cabs2(h11);
        cf32 a = cmulconj(r0, h00), b = cmulconj(h11, r1), cc = cmulconj(h10, r0), d = cmulconj(r1, h01);
        float t0r = a.r + b.r, t0i = a.i + b.i, t1r = d.r - cc.r, t1i = d.i - cc.i;
        if (rx == 0) { x0r = t0r; x0i = t0i; x1r = t1r; x1i = t1i; hh = hp; }
        else { x0r = x0r + t0r; x0i = x0i + t0i; x1r = x1r + t1r; x1i = x1i + t1i; hh = hh + hp; }
      }
      cf32 x0, x1;
      x0.r = x0r / hh * SQRT2F; x0.i = x0i / hh * SQRT2F; x1.r = x1r / hh * SQRT2F; x1.i = x1i / hh * SQRT2F;
      float w = hh * (fstd ? 2.0f : 1.0f) / chan_ref;
      emit(c, (int)g.qm[0], x0, w, inv_amp, g.cinit[0], idx, out0);
      emit(c, (int)g.qm[0], x1, w, inv_amp, g.cinit[0], idx + 1, out0);
      break;
    }
    default: {  // 2: closed-loop spatial multiplexing, 3: large-delay CDD
      if (g.nof_layers == 1) {
        float nr = 0, ni = 0, den = 0;
        for (int rx = 0; rx < A; rx++) {
          cf32 h0 = CE(0, rx, k), h1 = CE(1, rx, k), qh;
          switch (g.pmi) {
            case 0: qh = h1; break;
            case 1: qh.r = -h1.r; qh.i = -h1.i; break;
            case 2: qh.r = -h1.i; qh.i = h1.r; break;
            default: qh.r = h1.i; qh.i = -h1.r; break;
          }
          cf32 he; he.r = (h0.r + qh.r) * SQRT1_2F; he.i = (h0.i + qh.i) * SQRT1_2F;
          cf32 t = cmulconj(GRID(rx, k), he);
          float hp = cabs2(he);
          if (rx == 0) { nr = t.r; ni = t.i; den = hp; } else { nr = nr + t.r; ni = ni + t.i; den = den + hp; }
        }
        float dn = den + noise;
        cf32 x; x.r = nr / dn; x.i = ni / dn;
        emit(c, (int)g.qm[0], x, den * 2.0f / chan_ref, inv_amp, g.cinit[0], idx, out0);
      } else {
        float a = 0, d = 0, br = 0, bi = 0, z0r = 0, z0i = 0, z1r = 0, z1i = 0;
        for (int rx = 0; rx < A; rx++) {
          cf32 h0 = CE(0, rx, k), h1 = CE(1, rx, k), qh, y = GRID(rx, k);
          if (g.tx_scheme == 3) {
            if (idx & 1u) { qh.r = -h1.r; qh.i = -h1.i; } else qh = h1;
          } else if (g.pmi == 0) {
            qh = h1;
          } else {
            qh.r = -h1.i; qh.i = h1.r;
          }
          cf32 e0, e1;
          e0.r = (h0.r + qh.r) * 0.5f; e0.i = (h0.i + qh.i) * 0.5f;
          e1.r = (h0.r - qh.r) * 0.5f; e1.i = (h0.i - qh.i) * 0.5f;
          cf32 b = cmulconj(e1, e0);
          cf32 t0 = cmulconj(y, e0), t1 = cmulconj(y, e1);
          float p0 = cabs2(e0), p1 = cabs2(e1);
          if (rx == 0) { a = p0; d = p1; br = b.r; bi = b.i; z0r = t0.r; z0i = t0.i; z1r = t1.r; z1i = t1.i; }
          else { a = a + p0; d = d + p1; br = br + b.r; bi = bi + b.i; z0r = z0r + t0.r; z0i = z0i + t0.i; z1r = z1r + t1.r; z1i = z1i + t1.i; }
        }
        a = a + noise; d = d + noise;
        float det = a * d - (br * br + bi * bi);
        cf32 x0, x1;
        x0.r = (d * z0r - (br * z1r - bi * z1i)) / det;
        x0.i = (d * z0i - (br * z1i + bi * z1r)) / det;
        x1.r = (a * z1r - (br * z0r + bi * z0i)) / det;
        x1.i = (a * z1i - (br * z0i - bi * z0r)) / det;
        float w0 = det / d * 4.0f / chan_ref, w1 = det / a * 4.0f / chan_ref;
        if (g.qm[0]) emit(c, (int)g.qm[0], x0, w0, inv_amp, g.cinit[0], idx, out0);
        if (g.qm[1]) emit(c, (int)g.qm[1], x1, w1, inv_amp, g.cinit[1], idx, out1);
      }
      break;
    }
  }
#undef GRID
#undef CE
}
void lsn_launch_pdsch_demod(const LsnCellDev& c, const LsnGrantDev* g, const uint32_t* items, uint32_t nitems, const uint16_t* prefix, const cf32* grid,
                            const cf32* ce, const LsnChest* ch, int16_t* llr, hipStream_t s)
{
  if (nitems) LSN_LAUNCH(k_pdsch_demod, dim3(nitems, 14), dim3(192), 0, s, c, g, items, prefix, grid, ce, ch, llr);
}

// ------------------------------------------------------------------------------------------------ rate de-matching
// 36.212 5.1.4.1.2 inverted, as a kernel of its own in front of the turbo decoder: the gather over the circular buffer is
// memory-latency work that wants many resident wavefronts, the decoder is register-bound and runs one wavefront per SIMD.
// Workgroup per code block.  The block's rate-matched soft bits e[0 .. E) are staged in LDS with coalesced 16-byte loads;
// thread t then builds word t of the decoder's TRANSPOSED layout (t = (x % W) * P + x / W  <=>  x = (t % P) * W + t / P):
// the three streams of trellis position x are looked up through the closed-form rank of lsn_rm.h (first e index of a
// circular-buffer entry), repetitions e[rank + m * nn] are added, the sum is clipped to +-511 and the three 10-bit fields
// are packed (systematic | parity 1 << 10 | parity 2 << 20; filler bits are known zeros: -511).  Words K .. K+11 carry the
// twelve termination values (stream s, position K + j at K + 4 s + j) as int32.  Blocks whose E exceeds the staging
// area read e[] from global memory instead (same arithmetic).
#define RM_NT 256
// Round 6 (last session): the ranks come from ONE 8-byte LDS entry per buffer column (lsn_rm.h: lsn_rm_rank01_fast / lsn_rm_rank2_fast - the streams 0 and 1 of a
// position share their column), the geometry is filled by 32 lanes of the first wavefront instead of one thread, and without repetition (E <= nn: every
// block of the downlink workloads) a value is one LDS read at min(rank, E) - e[E] is staged as zero - instead of a loop behind a branch.
template <bool STAGED>
__device__ __forceinline__ int rm_sum(const int16_t* __restrict__ e, const int16_t* es, int rank, int E, int nn)
{
  int acc = 0;
  for (int k = rank; k < E; k += nn) acc += STAGED ? (int)es[k] : (int)e[k];
  return acc > LSN_LLR_CLIP ? LSN_LLR_CLIP : (acc < -LSN_LLR_CLIP ? -LSN_LLR_CLIP : acc);
}
__global__ __launch_bounds__(RM_NT) void k_rm(const LsnCbDev* __restrict__ cbs, const int16_t* __restrict__ llr, uint32_t* __restrict__ spp_g, uint32_t seg)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char rm_smem[];
  __shared__ LsnRmGeom geom;
  __shared__ LsnRmCol tab[32];
  int16_t* es = (int16_t*)rm_smem;
  const LsnCbDev cb = cbs[blockIdx.x];
  const int tid = threadIdx.x, K = (int)cb.K, F = (int)cb.F, E = (int)cb.E;
  const int P = cb.nwin ? (int)cb.nwin : lsn_turbo_nwin(K), W = K / P;
  const int ND = 32 * ((K + 4 + 31) >> 5) - (K + 4);
  const int16_t* e = llr + cb.e_off;  // e_off is a multiple of 8 entries (16 bytes) for the first block of a codeword only
  const bool staged = (uint32_t)E <= seg;
  if (tid < 64) {
    // geometry: lane c < 32 counts the <NULL>s of buffer column c, an inclusive shuffle scan gives the prefix sums (the lanes 32..63 carry zeros)
    int c01 = 0, f2 = 0, c2 = 0;
    if (tid < 32) lsn_rm_geom_col(ND, F, tid, &c01, &f2, &c2);
    int i01 = c01, i2 = c2;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int t = __shfl_up(i01, off), u = __shfl_up(i2, off);
      if (tid >= off) { i01 += t; i2 += u; }
    }
    if (tid < 32) { geom.cnt01[tid] = (uint8_t)c01; geom.first2[tid] = (uint8_t)f2; geom.pre01[tid] = i01 - c01; geom.pre2[tid] = i2 - c2; }
    if (tid == 31) { geom.pre01[32] = i01; geom.pre2[32] = i2; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tid == 0) lsn_rm_geom_finish(geom, K, F, (int)cb.rv);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (tid < 32) tab[tid] = lsn_rm_fast_col(geom, tid);
  }
  if (staged) {
    // 16-byte loads from the aligned address at or below e; the entry behind the last one, e[E], is staged as zero (what a rank >= E reads)
    const uintptr_t a0 = (uintptr_t)e & ~(uintptr_t)15;
    const int skew = (int)(((uintptr_t)e - a0) >> 1);  // entries between the aligned address and e[0]
    const uint4* src = (const uint4*)a0;
    const int end = skew + E, vend = end >> 3;
    for (int v = tid; v <= vend; v += RM_NT) {
      uint4 d = make_uint4(0u, 0u, 0u, 0u);
      if (8 * v < end) d = src[v];
      if (v == vend) {
        const int q = end & 7;
        const uint32_t keep = (q & 1) ? 0x0000FFFFu : 0u;   // the word of entry `end`: its low half stays when `end` is the high half
        switch (q >> 1) { case 0: d.x &= keep; break; case 1: d.y &= keep; break; case 2: d.z &= keep; break; default: d.w &= keep; break; }
      }
      ((uint4*)es)[v] = d;
    }
    es += skew;
  }
  __syncthreads();
  const int nn = geom.nn;
  uint32_t* out = spp_g + cb.spp_off;
  const bool single = staged && E <= nn;
  // the three streams of trellis position x
  auto streams = [&](int x, int& v0, int& v1, int& v2) {
    const int y = x + ND;
    int r0, r1;
    lsn_rm_rank01_fast(tab[lsn_rm_col_of(y)], nn, y, &r0, &r1);
    const int r2 = lsn_rm_rank2_fast(tab[lsn_rm_col_of(y - 1)], nn, y - 1);
    if (single) {
      v0 = (int)es[r0 < E ? r0 : E]; v1 = (int)es[r1 < E ? r1 : E]; v2 = (int)es[r2 < E ? r2 : E];
    } else if (staged) {
      v0 = rm_sum<true>(e, es, r0, E, nn); v1 = rm_sum<true>(e, es, r1, E, nn); v2 = rm_sum<true>(e, es, r2, E, nn);
    } else {
      v0 = rm_sum<false>(e, es, r0, E, nn); v1 = rm_sum<false>(e, es, r1, E, nn); v2 = rm_sum<false>(e, es, r2, E, nn);
    }
  };
  // word t of the transposed layout, t = tq * P + tr with tq, tr carried along from one pass of the workgroup to the next
  int tq = tid / P, tr = tid - tq * P;
  const int dq = RM_NT / P, dr = RM_NT - dq * P;
  for (int t = tid; t < K; t += RM_NT, tq += dq, tr += dr) {
    if (tr >= P) { tr -= P; tq++; }
    const int x = (int)__umul24((uint32_t)tr, (uint32_t)W) + tq;   // = (t % P) * W + t / P
    int v0, v1, v2;
    streams(x, v0, v1, v2);
    if (x < F) v0 = v1 = -LSN_LLR_CLIP;   // filler bits are known zeros
    out[t] = ((uint32_t)v0 & 0x3FFu) | (((uint32_t)v1 & 0x3FFu) << 10) | (((uint32_t)v2 & 0x3FFu) << 20);
  }
  // the twelve termination values: stream s, position K + j at word K + 4 s + j
  if (tid < 4) {
    int v0, v1, v2;
    streams(K + tid, v0, v1, v2);
    out[K + tid] = (uint32_t)v0; out[K + 4 + tid] = (uint32_t)v1; out[K + 8 + tid] = (uint32_t)v2;
  }
}
// the staging area is sized by the largest E of the launch, capped at 64 KiB (two workgroups per CU at least)
void lsn_launch_rm(const LsnCbDev* cb, const int16_t* llr, uint32_t* spp, uint32_t ncb, uint32_t emax, hipStream_t s)
{
  static std::atomic<uint64_t> attr_done{0};
  lsn_func_max_lds((const void*)k_rm, 65536, attr_done, "k_rm");
  const uint32_t cap = (65536 - 32) / 2;
  const uint32_t seg = emax < cap ? emax : cap;
  const size_t lds = (((size_t)seg + 16) * 2 + 15) & ~(size_t)15;  // + the skew in front of e[0] and the tail of the last 16-byte load
  if (ncb) LSN_LAUNCH(k_rm, dim3(ncb), dim3(RM_NT), lds, s, cb, llr, spp, seg);
}

// ------------------------------------------------------------------------------------------------ HARQ soft combining
// The soft buffer of a (RNTI, HARQ process, transport block) holds the de-rate-matched code blocks of the transmissions so far in the decoder's own
// packed format (k_rm's output: K words of three 10-bit fields + 12 termination values).  A retransmission is combined field by field,
// acc = clip(acc + cur, +-511), and decoded from the result; a new transmission replaces the buffer (srsran_softbuffer_rx_reset_tbs).
// Round 6: the combination never writes the pool.  Inside one commit turn the content of a buffer lives in one of three places - the pool (what earlier
// chunks left), the chunk's keep store (a failed new transmission of this chunk) or the turn's scratch area (an earlier combination of this chunk) - and
// every combination writes a NEW piece of the scratch area, so that a whole batch of retransmissions can be combined and decoded ahead of the sequential
// commit walk without touching anything the walk may still want in its old state (lsn_engine.cc: harqScout / harqRunBatch).  The pool is written by the
// copy form only, once per touched buffer at the end of the turn.
// One workgroup per code block.  LsnCbDev::reserved = where the accumulated values are read (place in the two top bits: 0 pool, 1 scratch, 2 keep store;
// word offset below), e_off = the words of the current transmission in the keep store (combine form), spp_off = where the result goes (combine: scratch,
// copy: pool).
__global__ __launch_bounds__(256) void k_harq_combine(const LsnCbDev* __restrict__ cbs, const uint32_t* __restrict__ keep, uint32_t* pool, uint32_t* scratch, uint32_t copy)
{
  const LsnCbDev cb = cbs[blockIdx.x];
  const uint32_t place = cb.reserved >> 30, off = cb.reserved & 0x3FFFFFFFu;
  const uint32_t* a = (place == 0u ? (const uint32_t*)pool : place == 1u ? (const uint32_t*)scratch : keep) + off;
  const int K = (int)cb.K;
  if (copy) {
    uint32_t* d = pool + cb.spp_off;
    for (int t = threadIdx.x; t < K + 12; t += 256) d[t] = a[t];
    return;
  }
  const uint32_t* c = keep + cb.e_off;
  uint32_t* d = scratch + cb.spp_off;
  auto clip = [](int v) { return v > LSN_LLR_CLIP ? LSN_LLR_CLIP : (v < -LSN_LLR_CLIP ? -LSN_LLR_CLIP : v); };
  for (int t = threadIdx.x; t < K + 12; t += 256) {
    const uint32_t w = c[t], o = a[t];
    if (t < K) {
      const int v0 = clip(((int)(w << 22) >> 22) + ((int)(o << 22) >> 22)), v1 = clip(((int)(w << 12) >> 22) + ((int)(o << 12) >> 22)), v2 = clip(((int)(w << 2) >> 22) + ((int)(o << 2) >> 22));
      d[t] = ((uint32_t)v0 & 0x3FFu) | (((uint32_t)v1 & 0x3FFu) << 10) | (((uint32_t)v2 & 0x3FFu) << 20);
    } else {
      d[t] = (uint32_t)clip((int)w + (int)o);
    }
  }
}
void lsn_launch_harq_combine(const LsnCbDev* cbs, uint32_t ncb, const uint32_t* keep, uint32_t* pool, uint32_t* scratch, bool copy, hipStream_t s)
{
  if (ncb) LSN_LAUNCH(k_harq_combine, dim3(ncb), dim3(256), 0, s, cbs, keep, pool, scratch, copy ? 1u : 0u);
}

// ------------------------------------------------------------------------------------------------ turbo decoder
// One workgroup per code block, thread = trellis window (P = lsn_turbo_nwin(K) windows of W = K/P steps).  k_turbo<128>: blocks of more than
// 64 windows (two working wavefronts) and the few blocks of <= 64 windows with K > 3072 (one working wavefront, the second leaves at once) -
// 40 KiB of LDS at most, four per CU; k_turbo<64>: everything else, <= 22 KiB.  Two wavefronts per SIMD (256 registers).
// LDS: spp[K] packs the three rate-dematched soft streams of a position (10-bit signed fields: systematic | parity 1 |
// parity 2), ext[K] holds extrinsic * 2 + hard decision.  Both are stored TRANSPOSED, idx(x) = (x % W) * P + x / W:
// the in-order decoder reads consecutive lanes = consecutive addresses and the QPP-interleaved one is (nearly)
// conflict free by the contention-free property of the QPP (36.212 5.1.3.2.3).
// Schedule per constituent decoder: forward sweep in sub-blocks of TB_S steps (operands of a sub-block are fetched
// from LDS in one burst, the recursion then runs on registers), alpha check-pointed at sub-block starts; backward
// sub-block by sub-block: burst fetch, recompute the TB_S alphas into registers, beta + LLR + extrinsic.
// The interleaver addresses come from a table in global memory (L2), fetched one sub-block ahead.
// Window-boundary metrics of the previous iteration (next-iteration initialisation) stay in registers and change lanes through the
// check-point area.
// The recursions run on packed int16 pairs: lsn_turbo_core.h holds the layouts, the word-length argument and the per-lane text, which
// tests/native/test_turbo_core.cc also runs on the CPU against the oracle's decoder.
#include "lsn_turbo_core.h"

// GF(2) polynomial product a*b mod g (24-bit CRC generators, poly includes the x^24 term)
__device__ __forceinline__ uint32_t mulmod24(uint32_t a, uint32_t b, uint32_t poly)
{
  uint32_t r = 0;
#pragma unroll
  for (int i = 23; i >= 0; i--) {
    r <<= 1;
    r ^= (r & 0x1000000u) ? poly : 0u;
    r ^= ((b >> i) & 1u) ? a : 0u;
  }
  return r & 0xFFFFFFu;
}

// Synchronisation among the threads that work on ONE code block: a block of more than 64 windows is decoded by both wavefronts of its workgroup
// (workgroup barrier); a block of at most 64 windows by one wavefront - its partner in the workgroup has left, or decodes ANOTHER block at its own
// pace (paired launch, k_turbo below), so a workgroup barrier would be wrong: the LDS operations of one wavefront complete in order, what is needed is
// that the compiler keeps them in order.
__device__ __forceinline__ void tb_sync(int nt)
{
  if (nt == 128) { __syncthreads(); return; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one constituent decoder over all windows (lane = window); nii_a / nii_b: boundary metrics (layout C) in registers
template <bool IL>
__device__ __forceinline__ void map_pass(const TurboLds& m, const uint32_t* il, int nt, int lane, bool active, int K, int P, int W,
                                         s2* nii_a, s2* nii_b, const s2* beta_tail)
{
  s2 a_end[4] = {s2{0, 0}, s2{0, 0}, s2{0, 0}, s2{0, 0}}, b_out[4] = {s2{0, 0}, s2{0, 0}, s2{0, 0}, s2{0, 0}};
  // lanes without a window sit the pass out (one exec-mask update per pass; rounds 2-4 let them shadow window 0 and paid a select per step for it)
  if (active) lsn_map_pass_lane<IL>(m, il, nt, lane, K, P, W, nii_a, nii_b, beta_tail, a_end, b_out);
  // next-iteration initialisation: window p starts from the end of window p-1 and ends at the start of window p+1.
  // The exchange goes through the (now idle) check-point area, slots 0 and 1.
  tb_sync(nt);
  lsn_ckpt_store(m.ckpt, nt, 0, lane, a_end, m.cw, m.ch);
  lsn_ckpt_store(m.ckpt, nt, 1, lane, b_out, m.cw, m.ch);
  tb_sync(nt);
  const int lm = lane > 0 ? lane - 1 : 0, lq = lane + 1 < nt ? lane + 1 : lane;
  lsn_ckpt_load(m.ckpt, nt, 0, lm, nii_a, m.cw, m.ch);
  lsn_ckpt_load(m.ckpt, nt, 1, lq, nii_b, m.cw, m.ch);
  tb_sync(nt);
}

__device__ __forceinline__ void tail_beta(const int* ts, const int* tp, int* beta)
{
  int b[8], bn[8];
  for (int S = 0; S < 8; S++) b[S] = S == 0 ? 0 : LSN_NEG_METRIC;
  for (int t = 2; t >= 0; t--) {
    for (int S = 0; S < 8; S++) {
      int s1 = (S >> 2) & 1, s2 = (S >> 1) & 1, s3 = S & 1;
      int u = s2 ^ s3, z = s1 ^ s3, Sn = (s1 << 1) | s2;
      bn[S] = b[Sn] + (u ? ts[t] : 0) + (z ? tp[t] : 0);
    }
    for (int S = 0; S < 8; S++) b[S] = bn[S];
  }
  for (int S = 7; S >= 0; S--) beta[S] = b[S] - b[0];
}

// XOR-reduce a value over the working threads of the workgroup (nt = 64: one wave; 128: two waves through LDS scratch)
__device__ __forceinline__ uint32_t wg_xor(uint32_t v, int16_t* scratch, int tid, int nt)
{
  for (int off = 32; off > 0; off >>= 1) v ^= __shfl_xor(v, off);
  if (nt == 64) return v;
  uint32_t* w = (uint32_t*)scratch;
  __syncthreads();
  if ((tid & 63) == 0) w[tid >> 6] = v;
  __syncthreads();
  v = w[0] ^ w[1];
  __syncthreads();
  return v;
}

// Two wavefronts per SIMD (256 registers; the 22 the allocator spills sit outside the trellis loops): measured + 4 % on the pipeline against
// the 284-register build at one wavefront per SIMD (profiles/r03_experiments.txt) - the second wave covers the LDS waits of the first
#ifndef TB_WAVES_ATTR
#define TB_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
// Launch layout (round 5).  A decoder workgroup holds one of the CU's four 40 KB LDS slots and two of its eight wavefront slots whatever it decodes, and
// the engine is bound by exactly those slots (profiles/r05_experiments.txt: three workgroups per CU instead of four cost 16 %).  Half of the code blocks
// of the metric's workload have at most 64 windows - ONE working wavefront - and rounds 2-4 gave each of them a whole slot with an idle second
// wavefront.  Now:
//   workgroups [0, nsolo):  one code block each (cbs[wg]): two working wavefronts when it has more than 64 windows, else the second leaves at once;
//   workgroups [nsolo, ..): TWO code blocks of at most 64 windows, one per wavefront (cbs[nsolo + 2 (wg - nsolo) + wave]), each with its own half of
//                           the workgroup's LDS (TurboLds: index biases) - the host pairs blocks of K <= LSN_TURBO_PAIR_KMAX so that two halves fit a slot.
// The two wavefronts of a pair never synchronise with each other (tb_sync).
template <int NT>
__global__ __launch_bounds__(NT) TB_WAVES_ATTR void k_turbo(const uint32_t* __restrict__ crc_tab_a, const uint32_t* __restrict__ crc_tab_b, const uint32_t* __restrict__ il_tab,
                                              const LsnCbDev* __restrict__ cbs, const uint32_t* __restrict__ spp_g,
                                              uint8_t* __restrict__ payload, LsnCbRes* res, uint32_t kmax, uint32_t nsolo, uint32_t ncb, uint32_t kmax_pair)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // phase cycle counters (LsnCbRes::cyc_*) only in instrumented builds (-DLSN_TURBO_CYCLES): the production kernel reads no clock
#ifdef LSN_TURBO_CYCLES
#define TB_CLOCK() clock64()
#else
#define TB_CLOCK() 0ll
#endif
  const long long tc0 = TB_CLOCK();
  const bool paired = NT == 128 && blockIdx.x >= nsolo;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform: in a scalar register, so that the index biases below are scalars too
  const uint32_t cbi = paired ? nsolo + 2u * (blockIdx.x - nsolo) + wave : blockIdx.x;
  if (cbi >= ncb) return;  // (the odd block of the pairs: its partner wavefront has nothing to decode)
  if (paired) kmax = kmax_pair;
  const LsnCbDev cb = cbs[cbi];
  // The transport block of this code block is already lost when its first code block (decoded by an EARLIER launch on this stream)
  // failed: nothing this block could decode would reach the record stream, so it is not decoded at all
  if (cb.dep != LSN_CB_NODEP && res[cb.dep].ok == 0u) {
    if ((paired ? (threadIdx.x & 63u) : threadIdx.x) == 0) { LsnCbRes r{}; res[cb.res_idx] = r; }
    return;
  }
  const int lane = paired ? (int)(threadIdx.x & 63u) : (int)threadIdx.x, K = (int)cb.K, F = (int)cb.F;
  const int P = cb.nwin ? (int)cb.nwin : lsn_turbo_nwin(K), W = K / P;
  const uint32_t magicW = ((1u << 20) + (uint32_t)W - 1u) / (uint32_t)W;
  // a block of at most 64 windows: one working wavefront (solo: the second wavefront leaves before the first barrier)
  const int nt = (NT == 128 && (paired || P <= 64)) ? 64 : NT;
  if (lane >= nt) return;
  const bool active = lane < P;
  const uint32_t* il = il_tab + cb.il_off;
  TurboLds m;
  // one block: spp[kmax] ext[kmax + 8] ckpt (ext[K] = spare slot for idle lanes); two blocks: spp0 spp1 (kmax + 8 words each) ext0 ext1 (kmax + 8 halves each)
  // ckpt0 ckpt1 - the second block is reached through index biases, the three pointers are the same for both wavefronts (TurboLds)
  const uint32_t kk = kmax + 8u;
  m.spp = (uint32_t*)smem; m.ext = (int16_t*)(m.spp + (paired ? 2u * kk : kmax)); m.ckpt = (uint8_t*)(m.ext + (paired ? 2u * kk : kk));
  m.bias = paired ? (int)(wave * kk) : 0; m.cw = paired ? (int)(wave * (uint32_t)(TB_CKPT_BYTES / 4)) : 0; m.ch = 2 * m.cw;
  // the check-point area doubles as scratch for the 12 termination values
  int* tail = (int*)(m.ckpt + 2048) + m.cw;
  // ---- soft data of the block: K packed words (already in the transposed layout) + 12 termination values, written by k_rm ----
  {
    const uint32_t* src = spp_g + cb.spp_off;  // 16-byte aligned, K is a multiple of 8
    for (int i = 4 * lane; i < K; i += 4 * nt) *(uint4*)&m.spp[i + m.bias] = *(const uint4*)&src[i];
    for (int i = 8 * lane; i < K; i += 8 * nt) *(uint4*)&m.ext[i + m.bias] = make_uint4(0u, 0u, 0u, 0u);
    if (lane < 12) tail[lane] = (int)src[K + lane];
  }
  tb_sync(nt);
  // ---- termination (36.212 5.1.3.2.2): tail[s*4 + j] = stream s at position K + j ----
  // The two termination vectors (layout C) are wanted by ONE lane - the last window - at the turn of every pass: they live in LDS behind the check-point
  // area(s), not in eight registers of every lane (round 6, last session: with the registers of the layout cycle the allocator had started to spill the
  // values that live across passes - 5 MB of scratch write-back per launch)
  s2* bt1 = (s2*)(m.ckpt + (paired ? 2u : 1u) * TB_CKPT_BYTES) + (paired ? 8u * wave : 0u);
  s2* bt2 = bt1 + 4;
  if (lane == 0) {
    const int *s4 = tail, *q1 = tail + 4, *q2 = tail + 8;
    int ts1[3] = {s4[0], q2[0], q1[1]}, tp1[3] = {q1[0], s4[1], q2[1]};
    int ts2[3] = {s4[2], q2[2], q1[3]}, tp2[3] = {q1[2], s4[3], q2[3]};
    int b8[8];
    s2 v[4];
    tail_beta(ts1, tp1, b8); lsn_pack_c(b8, v);
    for (int k = 0; k < 4; k++) bt1[k] = v[k];
    tail_beta(ts2, tp2, b8); lsn_pack_c(b8, v);
    for (int k = 0; k < 4; k++) bt2[k] = v[k];
  }
  tb_sync(nt);  // scratch is dead from here on: the area becomes the check-point store
  const long long tc1 = TB_CLOCK();
  const uint32_t poly = cb.crc_b ? 0x1800063u : 0x1864CFBu;
  // weight of this thread's window in the block polynomial: x^((P-1-window) W) mod g
  const uint32_t cw = active ? (cb.crc_b ? crc_tab_b : crc_tab_a)[(P - 1 - lane) * W] : 0u;
  s2 na1[4], nb1[4], na2[4], nb2[4];
#pragma unroll
  for (int s = 0; s < 4; s++) { na1[s] = s2{0, 0}; nb1[s] = s2{0, 0}; na2[s] = s2{0, 0}; nb2[s] = s2{0, 0}; }
  int it = 0;
  bool ok = false;
  while (it < (int)cb.max_iter && !ok) {
    map_pass<false>(m, il, nt, lane, active, K, P, W, na1, nb1, bt1);
    map_pass<true>(m, il, nt, lane, active, K, P, W, na2, nb2, bt2);
    it++;
    // CRC over all K decided bits == 0  <=>  data || parity divisible by g(x).  The thread sums, over its own window, the weights x^(K-1-x) mod g of the
    // positions whose decided bit is 1: it walks the window from its END, where the weight is cw, and multiplies the weight by x per step (rounds 1-5 ran a
    // Horner scheme over the window and multiplied the result by cw - 24 more shift / reduce / add steps per thread and iteration).  The decided bits are read
    // eight steps per burst: one LDS round trip per burst instead of one per step.  An XOR reduction over the windows follows.
    uint32_t rem = 0;
    if (active) {
      uint32_t q = cw;
      auto step = [&](uint32_t e) {
        uint32_t mk;   // all ones when the decided bit is 1 (stated as the instruction: the compiler turns every C form of it into and / compare / select)
        asm("v_bfe_i32 %0, %1, 0, 1" : "=v"(mk) : "v"(e));
        rem ^= mk & q;
        q <<= 1;
        q ^= (q & 0x1000000u) ? poly : 0u;
      };
      int t = W - 1;
      for (int k = W & 7; k > 0; k--, t--) step((uint32_t)m.ext[t * P + lane + m.bias]);   // (window lengths are 32 .. 95: the W mod 8 steps at the end first)
      for (; t >= 7; t -= 8) {
        uint32_t e[8];
#pragma unroll
        for (int k = 0; k < 8; k++) e[k] = (uint32_t)m.ext[(t - k) * P + lane + m.bias];
#pragma unroll
        for (int k = 0; k < 8; k++) step(e[k]);
      }
    }
    ok = wg_xor(rem, (int16_t*)m.ckpt, lane, nt) == 0;
  }
  const int it_run = it;
  const long long tc2 = TB_CLOCK();
  // ---- output: payload bytes of this code block + its CRC24A remainder contribution (each thread a contiguous run) ----
  const int nout = (int)cb.out_bytes;
  uint8_t* outp = payload + cb.out_off;
  const int per = (nout + nt - 1) / nt, j0 = lane * per, j1 = (j0 + per < nout) ? j0 + per : nout;
  uint32_t rema = 0;
  for (int j = j0; j < j1; j++) {
    uint32_t byte = 0, e[8];
#pragma unroll
    for (int q = 0; q < 8; q++) e[q] = (uint32_t)m.ext[tr_idx(F + 8 * j + q, W, P, magicW) + m.bias];   // the eight reads of a byte in one burst
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const uint32_t bit = e[q] & 1u;
      byte = (byte << 1) | bit;
      rema = (rema << 1) | bit;
      rema ^= (rema & 0x1000000u) ? 0x1864CFBu : 0u;
    }
    outp[j] = (uint8_t)byte;
  }
  if (j0 < j1) rema = mulmod24(rema, crc_tab_a[8 * (nout - j1)], 0x1864CFBu);
  rema = wg_xor(rema, (int16_t*)m.ckpt, lane, nt);
  if (lane == 0) {
    const long long tc3 = TB_CLOCK();
    LsnCbRes r; r.ok = ok ? 1u : 0u; r.iters = (uint32_t)it; r.rem_a = rema; r.iters_run = (uint32_t)it_run;
    r.cyc_rm = (uint32_t)(tc1 - tc0); r.cyc_map = (uint32_t)(tc2 - tc1); r.cyc_out = (uint32_t)(tc3 - tc2); r.cyc_all = (uint32_t)(tc3 - tc0);
    res[cb.res_idx] = r;
  }
}

size_t lsn_turbo_lds_bytes(uint32_t kmax) { return 6 * (size_t)kmax + 16 + TB_CKPT_BYTES + 64; }   // spp ext check-points termination vectors
static size_t turbo_lds_bytes_nt(uint32_t kmax, int) { return lsn_turbo_lds_bytes(kmax); }

// Classic form (uplink, HARQ re-decodes): cb[0 .. n128) in two-wavefront workgroups, cb[n128 .. n128 + n64) in one-wavefront workgroups; each range is
// launched with the LDS size of its largest block (40 KiB at K = 6144 -> four code blocks per CU)
void lsn_launch_turbo(const LsnCellDev& c, const LsnCbDev* cb, const uint32_t* spp, uint8_t* payload, LsnCbRes* res, uint32_t n128, uint32_t kmax128,
                      uint32_t n64, uint32_t kmax64, hipStream_t s, hipEvent_t between)
{
  lsn_launch_turbo_packed(c, cb, spp, payload, res, n128, kmax128, 0, 0, s);
  if (between) (void)hipEventRecord(between, s);
  static std::atomic<uint64_t> attr64{0};
  lsn_func_max_lds((const void*)k_turbo<64>, (int)turbo_lds_bytes_nt(6144, 64), attr64, "k_turbo<64>");
  auto fix = [](uint32_t k) { return ((k < 512 ? 512u : k) + 7u) & ~7u; };  // the scratch in the check-point area needs room
  if (n64) LSN_LAUNCH(k_turbo<64>, dim3(n64), dim3(64), turbo_lds_bytes_nt(fix(kmax64), 64), s, c.crc_tab_a, c.crc_tab_b, c.turbo_il, cb + n128, spp, payload, res, fix(kmax64), n64, n64, 0u);
}

// One launch for a whole decode phase: cb[0 .. nsolo) one block per workgroup, cb[nsolo .. nsolo + npair) two blocks (K <= LSN_TURBO_PAIR_KMAX, at most 64
// windows each) per workgroup - see k_turbo
void lsn_launch_turbo_packed(const LsnCellDev& c, const LsnCbDev* cb, const uint32_t* spp, uint8_t* payload, LsnCbRes* res, uint32_t nsolo, uint32_t kmax_solo,
                             uint32_t npair, uint32_t kmax_pair, hipStream_t s)
{
  static std::atomic<uint64_t> attr128{0};
  lsn_func_max_lds((const void*)k_turbo<128>, (int)turbo_lds_bytes_nt(6144, 128), attr128, "k_turbo<128>");
  auto fix = [](uint32_t k) { return ((k < 512 ? 512u : k) + 7u) & ~7u; };  // the scratch in the check-point area needs room
  if (!nsolo && !npair) return;
  const uint32_t ks = nsolo ? fix(kmax_solo) : 512u, kp = npair ? fix(kmax_pair) : 512u;
  const size_t pair_lds = npair ? 12 * ((size_t)kp + 8) + 2 * TB_CKPT_BYTES + 64 : 0;   // spp0 spp1 ext0 ext1 ckpt0 ckpt1 bt0 bt1 (k_turbo)
  const size_t lds = std::max(nsolo ? turbo_lds_bytes_nt(ks, 128) : (size_t)0, pair_lds);
  LSN_LAUNCH(k_turbo<128>, dim3(nsolo + (npair + 1) / 2), dim3(128), lds, s, c.crc_tab_a, c.crc_tab_b, c.turbo_il, cb, spp, payload, res, ks, nsolo, nsolo + npair, kp);
}
