// Do kernels on two HIP streams overlap on this stack?  A spin kernel (busy-waits `cycles` of s_memtime) with a chosen grid, block size
// and dynamic LDS is launched on one stream twice (serial) and on two streams once each; wall time of the pair says whether the
// second ran beside the first.  Cases: small blocks without LDS; 1024-thread blocks with 128 KB of LDS on 84 CUs each (the shape of a
// half-batch one-round K-group GEMM launch); one 168-block launch of that kind beside a small kernel.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_concurrency.hip -o tools/probe_concurrency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void spin(long cycles, int* sink) {
  extern __shared__ char lds[];
  const long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
  if (threadIdx.x == 0 && sink) sink[blockIdx.x] = lds[0];
}

static double run(hipStream_t a, hipStream_t b, int grid, int block, size_t lds, long cycles, int grid_b, int block_b, size_t lds_b) {
  hipDeviceSynchronize();
  auto t0 = std::chrono::high_resolution_clock::now();
  hipLaunchKernelGGL(spin, dim3(grid), dim3(block), lds, a, cycles, nullptr);
  hipLaunchKernelGGL(spin, dim3(grid_b), dim3(block_b), lds_b, b, cycles, nullptr);
  hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
}

int main() {
  hipStream_t s1, s2;
  hipStreamCreate(&s1);
  hipStreamCreate(&s2);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const long cyc = 1000000;  // readcyclecounter ticks (a few hundred microseconds)
  struct { const char* name; int g, b; size_t l; int g2, b2; size_t l2; } cases[] = {
      {"64 blocks x 256 threads, no LDS  |  same", 64, 256, 0, 64, 256, 0},
      {"84 blocks x 1024 threads, 128 KB LDS  |  same", 84, 1024, 131072, 84, 1024, 131072},
      {"168 blocks x 1024 threads, 128 KB LDS  |  64 blocks x 256 threads", 168, 1024, 131072, 64, 256, 0},
      {"256 blocks x 1024 threads, 128 KB LDS  |  64 blocks x 256 threads", 256, 1024, 131072, 64, 256, 0},
  };
  for (auto& c : cases) {
    run(s1, s1, c.g, c.b, c.l, cyc, c.g2, c.b2, c.l2);
    double ser = 0, par = 0;
    for (int i = 0; i < 5; ++i) { ser += run(s1, s1, c.g, c.b, c.l, cyc, c.g2, c.b2, c.l2); par += run(s1, s2, c.g, c.b, c.l, cyc, c.g2, c.b2, c.l2); }
    printf("%-75s one stream %7.1f us   two streams %7.1f us\n", c.name, ser / 5, par / 5);
  }
  return 0;
}
