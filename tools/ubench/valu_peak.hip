// Micro-benchmark: PEAK issue rate of plain wave64 VALU instructions on gfx950 - the denominator of the "VALU utilisation" figures in
// bench.py / DESIGN.md.  16 independent dependency chains per wave, 256 VALU instructions per loop iteration and nothing else vector in
// the loop (the loop counter is scalar), 1 / 2 / 4 / 8 waves per SIMD on every CU.  The guide's figure is one wave64 instruction per 2
// cycles per SIMD-32 (MI355X_MICROARCH.md "Wave scheduling"): 256 CUs x 4 SIMDs x 2.4 GHz / 2 = 1228.8 G wave-instructions/s.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_peak valu_peak.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define N_IT 4096
#define R16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define R16x16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X) R16(X)
template <int WHICH>
__global__ __launch_bounds__(512) void k(uint32_t* out, uint32_t seed)
{
  uint32_t a[16];
  float f[16];
  const uint32_t b = seed * 2654435761u + threadIdx.x;
  const float fb = 1.0f + 1e-7f * (float)threadIdx.x, fc = 1e-9f * (float)seed;
  for (int i = 0; i < 16; i++) { a[i] = seed + i * 77u + threadIdx.x; f[i] = (float)i + fb; }
  for (int it = 0; it < N_IT; it++) {
#define OP_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAX(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[i]) : "v"(fb), "v"(fc));
#define OP_PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAX3(i) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(seed));
#define OP_PKMAX(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_SUB(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_LSHL(i) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
#define OP_ASHR(i) asm volatile("v_ashrrev_i32 %0, 3, %0" : "+v"(a[i]));
#define OP_BFE(i) asm volatile("v_bfe_i32 %0, %0, 10, 10" : "+v"(a[i]));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : );
#define OP_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a[i]) : "v"(b));
#define OP_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ACCW(i) asm volatile("v_accvgpr_write_b32 a" #i ", %0" : : "v"(a[i]));
#define OP_ACCR(i) asm volatile("v_accvgpr_read_b32 %0, a" #i : "=v"(a[i]));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
#define OP_MIN(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_MAXI16(i) asm volatile("v_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define OP_ADDMAX(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b)); asm volatile("v_max_i32 %0, %0, %1" : "+v"(a[(i + 8) & 15]) : "v"(b));
#define OP_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    if (WHICH == 0) { R16x16(OP_ADD) }
    if (WHICH == 1) { R16x16(OP_MAX) }
    if (WHICH == 2) { R16x16(OP_FMA) }
    if (WHICH == 3) { R16x16(OP_PKADD) }
    if (WHICH == 4) { R16x16(OP_ADD3) }
    if (WHICH == 5) { R16x16(OP_MAX3) }
    if (WHICH == 6) { R16x16(OP_PKMAX) }
    if (WHICH == 7) { R16x16(OP_SUB) }
    if (WHICH == 8) { R16x16(OP_AND) }
    if (WHICH == 9) { R16x16(OP_LSHL) }
    if (WHICH == 10) { R16x16(OP_ASHR) }
    if (WHICH == 11) { R16x16(OP_BFE) }
    if (WHICH == 12) { R16x16(OP_CNDMASK) }
    if (WHICH == 13) { R16x16(OP_MAD24) }
    if (WHICH == 14) { R16x16(OP_MUL24) }
    if (WHICH == 15) { R16x16(OP_ACCW) }
    if (WHICH == 16) { R16x16(OP_ACCR) }
    if (WHICH == 17) { R16x16(OP_MOV) }
    if (WHICH == 18) { R16x16(OP_MIN) }
    if (WHICH == 19) { R16x16(OP_MAXI16) }
    if (WHICH == 20) { R16(OP_ADDMAX) R16(OP_ADDMAX) R16(OP_ADDMAX) R16(OP_ADDMAX) R16(OP_ADDMAX) R16(OP_ADDMAX) R16(OP_ADDMAX) R16(OP_ADDMAX) }
    if (WHICH == 21) { R16x16(OP_XOR) }
  }
  uint32_t r = 0;
  for (int i = 0; i < 16; i++) r ^= a[i] ^ __float_as_uint(f[i]);
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int W>
static void run(const char* name, int waves_per_cu)
{
  uint32_t* out;
  hipMalloc(&out, 4 * 256 * 2048);
  const int blocks = 256, wpb = waves_per_cu > 8 ? 8 : waves_per_cu, bpc = waves_per_cu / wpb;  // workgroups of <= 8 waves, bpc of them per CU
  hipLaunchKernelGGL(k<W>, dim3(blocks * bpc), dim3(64 * wpb), 0, 0, out, 3u);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<W>, dim3(blocks * bpc), dim3(64 * wpb), 0, 0, out, 5u);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * waves_per_cu * N_IT * 256.0;
  const double g = insts / (ms * 1e6);
  printf("%-14s %2d waves/SIMD: %8.3f ms  %8.1f G wave-instructions/s chip-wide = %.2f cycles per instruction per SIMD at 2.4 GHz (%.1f %% of 1228.8 G)\n", name,
         waves_per_cu / 4, ms, g, 1024.0 * 2.4 / g, 100.0 * g / 1228.8);
  hipFree(out);
}
#define ALL(W) run<0>("v_add_u32", W); run<1>("v_max_i32", W); run<2>("v_fma_f32", W); run<3>("v_pk_add_u16", W); run<4>("v_add3_u32", W); run<5>("v_max3_i32", W); \
  run<6>("v_pk_max_i16", W); run<7>("v_sub_u32", W); run<8>("v_and_b32", W); run<9>("v_lshlrev_b32", W); run<10>("v_ashrrev_i32", W); run<11>("v_bfe_i32", W); \
  run<12>("v_cndmask_b32", W); run<13>("v_mad_u32_u24", W); run<14>("v_mul_u32_u24", W); run<15>("v_accvgpr_write", W); run<16>("v_accvgpr_read", W); run<17>("v_mov_b32", W); \
  run<18>("v_min_u32", W); run<19>("v_max_i16", W); run<20>("add+max pairs", W); run<21>("v_xor_b32", W);
int main()
{
  ALL(4) ALL(8) ALL(16)
  return 0;
}
