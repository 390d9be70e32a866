// Micro-benchmark: issue cost (cycles per wave instruction, one wave per SIMD) of the integer VALU ops the turbo / Viterbi
// kernels are built from, gfx950.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short s2 __attribute__((ext_vector_type(2)));
#define N_IT 16384
#define BODY8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)
template <int WHICH>
__global__ void k(uint32_t* out, long long* cyc, uint32_t seed)
{
  uint32_t a[8], b = seed * 2654435761u + threadIdx.x;
  for (int i = 0; i < 8; i++) a[i] = seed + i * 77u + threadIdx.x;
  long long t0 = clock64();
  for (int it = 0; it < N_IT; it++) {
#define OP_ADD(i) a[i] = a[i] + b;
#define OP_MAX(i) a[i] = (uint32_t)max((int)a[i], (int)b);
#define OP_PKADD(i) a[i] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(s2, a[i]) + __builtin_bit_cast(s2, b));
#define OP_PKMAX(i) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, b)));
#define OP_PKADDSEL(i) a[i] = __builtin_bit_cast(uint32_t, __builtin_shufflevector(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, a[i]), 1, 1) + __builtin_bit_cast(s2, b));
#define OP_MAX3(i) a[i] = (uint32_t)max(max((int)a[i], (int)b), (int)seed);
#define OP_ADD3(i) a[i] = a[i] + b + seed;
#define OP_PERM(i) a[i] = __builtin_amdgcn_perm(a[i], b, 0x05040100u);
#define OP_MUL24(i) a[i] = __umul24(a[i], b);
#define OP_MULLO(i) a[i] = a[i] * b;
#define OP_PKSAT(i) a[i] = __builtin_bit_cast(uint32_t, __builtin_elementwise_add_sat(__builtin_bit_cast(s2, a[i]), __builtin_bit_cast(s2, b)));
#define OP_BFE(i) a[i] = (uint32_t)(((int)(a[i] << 12)) >> 22) + 1u;
#define OP_BPERM(i) a[i] = (uint32_t)__shfl((int)a[i], (int)(b & 63));
#define OP_DPP(i) a[i] = a[i] + (uint32_t)__builtin_amdgcn_mov_dpp((int)a[i], 0x111, 0xf, 0xf, false);
    if (WHICH == 0) { BODY8(OP_ADD) }
    if (WHICH == 1) { BODY8(OP_MAX) }
    if (WHICH == 2) { BODY8(OP_PKADD) }
    if (WHICH == 3) { BODY8(OP_PKMAX) }
    if (WHICH == 4) { BODY8(OP_PKADDSEL) }
    if (WHICH == 5) { BODY8(OP_MAX3) }
    if (WHICH == 6) { BODY8(OP_ADD3) }
    if (WHICH == 7) { BODY8(OP_PERM) }
    if (WHICH == 8) { BODY8(OP_MUL24) }
    if (WHICH == 9) { BODY8(OP_MULLO) }
    if (WHICH == 10) { BODY8(OP_PKSAT) }
    if (WHICH == 11) { BODY8(OP_BFE) }
    if (WHICH == 12) { BODY8(OP_BPERM) }
    if (WHICH == 13) { BODY8(OP_DPP) }
    b = b * 1664525u + 1013904223u;
  }
  long long t1 = clock64();
  uint32_t r = 0;
  for (int i = 0; i < 8; i++) r ^= a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int W>
static void run(const char* name, int waves_per_block, int instr_per_op)
{
  uint32_t* out; long long* cyc;
  hipMalloc(&out, 4 * 256 * 1024); hipMalloc(&cyc, 8 * 1024);
  const int blocks = 256;  // one block per CU
  hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 3u);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 5u);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < blocks; i++) avg += (double)h[i]; avg /= blocks;
  printf("%-28s waves/CU %2d: %7.2f ticks per wave-op | kernel %.3f ms -> %.2f ns per wave-op per wave, tick = %.3f ns, chip %.1f G wave-ops/s (%d instr/op)\n", name,
         waves_per_block, avg / (N_IT * 8.0), ms, ms * 1e6 / (N_IT * 8.0), ms * 1e6 / avg, blocks * waves_per_block * N_IT * 8.0 / (ms * 1e6), instr_per_op);
  hipFree(out); hipFree(cyc);
}
int main()
{
  for (int w : {1, 4, 8, 16}) {
    printf("--- %d wave(s) per workgroup (= %s per SIMD), one workgroup per CU ---\n", w, w == 1 ? "1 on one SIMD" : w == 4 ? "1" : w == 8 ? "2" : "4");
#define R(W, NAME, N) if (w == 1) run<W>(NAME, 1, N); else if (w == 4) run<W>(NAME, 4, N); else if (w == 8) run<W>(NAME, 8, N); else run<W>(NAME, 16, N);
    R(0, "v_add_u32", 1) R(1, "v_max_i32", 1) R(2, "v_pk_add_u16", 1) R(3, "v_pk_max_i16", 1) R(4, "v_pk_add_u16 op_sel", 1) R(5, "v_max3_i32", 1)
    R(6, "v_add3_u32", 1) R(7, "v_perm_b32", 1) R(8, "v_mul_u32_u24", 1) R(9, "v_mul_lo_u32", 1) R(10, "v_pk_add_i16 clamp", 1) R(11, "shl+ashr+add", 3)
    R(12, "ds_bpermute_b32", 1) R(13, "v_mov_dpp+add", 2)
  }
  return 0;
}
