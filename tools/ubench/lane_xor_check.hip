#include <hip/hip_runtime.h>
__global__ void k(int* out)
{
  int m = threadIdx.x * 7 + 3;
  // xor 1, 2 via quad_perm; xor 4 = half_mirror + quad reverse; xor 8 = row_mirror + half_mirror
  int x1 = __builtin_amdgcn_mov_dpp(m, 0xB1, 0xf, 0xf, false);  // quad_perm [1,0,3,2]
  int x2 = __builtin_amdgcn_mov_dpp(m, 0x4E, 0xf, 0xf, false);  // quad_perm [2,3,0,1]
  int h = __builtin_amdgcn_mov_dpp(m, 0x141, 0xf, 0xf, false);  // row_half_mirror
  int x4 = __builtin_amdgcn_mov_dpp(h, 0x1B, 0xf, 0xf, false);  // quad_perm [3,2,1,0]
  int r = __builtin_amdgcn_mov_dpp(m, 0x140, 0xf, 0xf, false);  // row_mirror
  int x8 = __builtin_amdgcn_mov_dpp(r, 0x141, 0xf, 0xf, false);
  auto p16 = __builtin_amdgcn_permlane16_swap(m, m, false, false);
  auto p32 = __builtin_amdgcn_permlane32_swap(m, m, false, false);
  int x16 = (threadIdx.x & 16) ? p16[0] : p16[1];
  int x32 = (threadIdx.x & 32) ? p32[0] : p32[1];
  out[threadIdx.x * 6 + 0] = x1; out[threadIdx.x * 6 + 1] = x2; out[threadIdx.x * 6 + 2] = x4; out[threadIdx.x * 6 + 3] = x8;
  out[threadIdx.x * 6 + 4] = x16; out[threadIdx.x * 6 + 5] = x32;
}
int main()
{
  int* d; hipMalloc(&d, 64 * 6 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[64 * 6]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++)
    for (int q = 0; q < 6; q++) {
      int want = (l ^ (1 << q)) * 7 + 3;
      if (h[l * 6 + q] != want) { if (bad < 10) printf("lane %d xor %d: got %d want %d\n", l, 1 << q, h[l * 6 + q], want); bad++; }
    }
  printf("bad = %d\n", bad);
  return bad != 0;
}
