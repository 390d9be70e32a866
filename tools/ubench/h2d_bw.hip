// tools/ubench/h2d_bw.hip - what the host -> GPU link of this box delivers, measured three ways (the ingest ceiling of lsn_phy_process_host /
// lsn_phy_process_file): (1) hipMemcpyAsync from pinned memory on 1 / 2 / 4 streams, (2) a kernel that reads the pinned host buffer
// directly (zero copy: what a k_ofdm fed from host memory would see), (3) both for several block sizes.  Build: hipcc --offload-arch=gfx950 -O2.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// every thread moves 16 B per iteration; consecutive threads read consecutive float4s (a wavefront = 1 KiB contiguous)
__global__ void k_pull(const float4* __restrict__ src, float4* __restrict__ dst, size_t n)
{
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}

// a compute hog: every CU busy with dependent integer work and some LDS traffic for `iters` rounds (stands in for the decode kernels)
__global__ void k_busy(uint32_t* out, int iters)
{
  __shared__ uint32_t sm[1024];
  uint32_t a = threadIdx.x + blockIdx.x, b = 0x9E3779B9u;
  sm[threadIdx.x] = a;
  __syncthreads();
  for (int i = 0; i < iters; i++) {
    a = a * 1664525u + 1013904223u; b ^= a >> 7; b += sm[(a >> 10) & 1023];
    if ((i & 63) == 0) { sm[threadIdx.x] = b; __syncthreads(); }
  }
  if (a == 0x12345u) out[0] = b;
}

int main(int argc, char** argv)
{
  const size_t MB = (size_t)1 << 20;
  const size_t total = (argc > 1 ? (size_t)atol(argv[1]) : 1536) * MB;
  int dev = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, dev));
  char bus[64];
  snprintf(bus, sizeof(bus), "%04x:%02x:%02x.0", pr.pciDomainID, pr.pciBusID, pr.pciDeviceID);
  printf("device %s  pci %s\n", pr.name, bus);
  for (const char* f : {"current_link_speed", "current_link_width", "max_link_speed", "max_link_width", "numa_node"}) {
    char path[256], buf[128] = {0};
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/%s", bus, f);
    FILE* fp = fopen(path, "r");
    if (fp) { if (fgets(buf, sizeof(buf), fp)) { buf[strcspn(buf, "\n")] = 0; printf("  %s = %s\n", f, buf); } fclose(fp); }
  }
  void *h = nullptr, *d = nullptr;
  for (int variant = 0; variant < 2; variant++) {
    const unsigned flags = variant == 0 ? hipHostMallocDefault : (hipHostMallocNonCoherent | hipHostMallocNumaUser);
    if (hipHostMalloc(&h, total, flags) != hipSuccess) { (void)hipGetLastError(); printf("hipHostMalloc flags %u refused\n", flags); continue; }
    memset(h, 1, total);
    CK(hipMalloc(&d, total));
    printf("pinned buffer %zu MB, hipHostMalloc flags 0x%x\n", total / MB, flags);
    hipStream_t st[8];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    for (size_t blk : {(size_t)32 * MB, (size_t)128 * MB, (size_t)384 * MB}) {
      for (int ns : {1, 2, 4, 8}) {
        // `ns` streams, each copying its own blocks back to back
        CK(hipDeviceSynchronize());
        const double t0 = now();
        size_t off = 0; int k = 0;
        while (off + blk <= total) { CK(hipMemcpyAsync((char*)d + off, (char*)h + off, blk, hipMemcpyHostToDevice, st[k % ns])); off += blk; k++; }
        CK(hipDeviceSynchronize());
        const double dt = now() - t0;
        printf("  memcpyAsync  block %4zu MB  streams %d : %6.2f GB/s\n", blk / MB, ns, off / dt / 1e9);
      }
      // one copy split over ns streams at a finer grain (4 MB pieces round-robin)
      for (int ns : {2, 4}) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        size_t off = 0; int k = 0; const size_t piece = 4 * MB;
        while (off + piece <= total) { CK(hipMemcpyAsync((char*)d + off, (char*)h + off, piece, hipMemcpyHostToDevice, st[k % ns])); off += piece; k++; }
        CK(hipDeviceSynchronize());
        printf("  memcpyAsync  4 MB pieces   streams %d : %6.2f GB/s\n", ns, off / (now() - t0) / 1e9);
        break;
      }
    }
    for (int blocks : {256, 1024, 4096, 16384}) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, st[0], (const float4*)h, (float4*)d, total / 16);
        CK(hipStreamSynchronize(st[0]));
        const double dt = now() - t0;
        if (rep) printf("  zero-copy kernel read  %5d blocks x 256 : %6.2f GB/s\n", blocks, total / dt / 1e9);
      }
    }
    // kernel pull and SDMA copy at the same time (do the two paths add up?)
    {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      const size_t half = total / 2;
      hipLaunchKernelGGL(k_pull, dim3(4096), dim3(256), 0, st[0], (const float4*)h, (float4*)d, half / 16);
      CK(hipMemcpyAsync((char*)d + half, (char*)h + half, half, hipMemcpyHostToDevice, st[1]));
      CK(hipDeviceSynchronize());
      printf("  kernel pull (half) + memcpyAsync (half) concurrently : %6.2f GB/s\n", total / (now() - t0) / 1e9);
    }
    // the same copies while the compute units are busy (what the pipeline sees: copies of the next blocks under the decode kernels)
    {
      uint32_t* dummy = nullptr;
      CK(hipMalloc((void**)&dummy, 64));
      for (int mode = 0; mode < 3; mode++) {
        CK(hipDeviceSynchronize());
        const int hog_blocks = mode == 2 ? 256 * 2 : 256 * 8;
        hipLaunchKernelGGL(k_busy, dim3(hog_blocks), dim3(256), 0, st[2], dummy, 3000000);   // several hundred ms of busy CUs
        const double t0 = now();
        if (mode == 0 || mode == 2) {
          size_t off = 0; const size_t blk = (size_t)384 * MB;
          while (off + blk <= total) { CK(hipMemcpyAsync((char*)d + off, (char*)h + off, blk, hipMemcpyHostToDevice, st[0])); off += blk; }
          CK(hipStreamSynchronize(st[0]));
          printf("  memcpyAsync 384 MB blocks under a compute hog (%d workgroups) : %6.2f GB/s\n", hog_blocks, off / (now() - t0) / 1e9);
        } else {
          hipLaunchKernelGGL(k_pull, dim3(256), dim3(256), 0, st[0], (const float4*)h, (float4*)d, total / 16);
          CK(hipStreamSynchronize(st[0]));
          printf("  zero-copy kernel read (256 workgroups) under a compute hog : %6.2f GB/s\n", total / (now() - t0) / 1e9);
        }
        const double t1 = now();
        CK(hipDeviceSynchronize());
        printf("    (hog ran %.0f ms beyond the copy)\n", (now() - t1) * 1e3);
      }
      // a D2H stream of small blocks next to the H2D copies (candidate tables / payloads go back while the next block comes in)
      {
        CK(hipDeviceSynchronize());
        const double t0 = now();
        size_t off = 0; const size_t blk = (size_t)384 * MB;
        while (off + blk <= total) {
          CK(hipMemcpyAsync((char*)d + off, (char*)h + off, blk, hipMemcpyHostToDevice, st[0]));
          for (int q = 0; q < 4; q++) CK(hipMemcpyAsync((char*)h + total - (q + 1) * 8 * MB, (char*)d + q * 8 * MB, 8 * MB, hipMemcpyDeviceToHost, st[1]));
          off += blk;
        }
        CK(hipStreamSynchronize(st[0]));
        printf("  memcpyAsync 384 MB blocks with 4 x 8 MB D2H per block on another stream : %6.2f GB/s\n", off / (now() - t0) / 1e9);
        CK(hipDeviceSynchronize());
      }
      CK(hipFree(dummy));
    }
    // device -> host for completeness (candidate tables and payloads go this way)
    {
      CK(hipDeviceSynchronize());
      const double t0 = now();
      CK(hipMemcpyAsync(h, d, total, hipMemcpyDeviceToHost, st[0]));
      CK(hipDeviceSynchronize());
      printf("  memcpyAsync D2H 1 stream : %6.2f GB/s\n", total / (now() - t0) / 1e9);
    }
    for (auto& s : st) CK(hipStreamDestroy(s));
    CK(hipFree(d)); CK(hipHostFree(h));
  }
  return 0;
}
