// LDS pipe micro-benchmark (gfx950): bytes per clock and CU of ds_write_b128 / b64 / b32 and ds_read_b128 / b64, lane-linear
// (conflict-free) addresses, 4 or 8 waves per CU.  Build: hipcc --offload-arch=gfx950 -O3 tools/bench_lds_rw.hip -o tools/bench_lds_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int MODE>  // 0 write b128, 1 write b64, 2 write b32, 3 read b128, 4 read b64
__global__ void k(int iters, long long* cyc, uint32_t* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  u32x4 v = {(uint32_t)tid, 1u, 2u, 3u};
  uint32_t acc = 0;
  const uint32_t base = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    u32x4 r4[8];
    u32x2 r2[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(base + (u * nthr + tid) * 16), "v"(v) : "memory");
      if (MODE == 1) asm volatile("ds_write_b64 %0, %1" ::"v"(base + (u * nthr + tid) * 8), "v"(u32x2{v.x, v.y}) : "memory");
      if (MODE == 2) asm volatile("ds_write_b32 %0, %1" ::"v"(base + (u * nthr + tid) * 4), "v"(v.x) : "memory");
      if (MODE == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(r4[u]) : "v"(base + (u * nthr + tid) * 16) : "memory");
      if (MODE == 4) asm volatile("ds_read_b64 %0, %1" : "=v"(r2[u]) : "v"(base + (u * nthr + tid) * 8) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 3) acc += r4[u].x + r4[u].w;
      if (MODE == 4) acc += r2[u].x + r2[u].y;
    }
  }
  __syncthreads();
  const long long t1 = clock64();
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
void run(const char* name, int bytes_per_lane, int threads) {
  long long* cyc; uint32_t* sink;
  hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
  const int iters = 2000;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, iters, cyc, sink);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 65536, 0, iters, cyc, sink);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  const double bytes = (double)iters * 8 * threads * bytes_per_lane;
  // clock64 = s_memtime (100 MHz constant on gfx9?) -> report by wall time at an assumed 2.4 GHz too
  printf("%-14s %4d thr: %7.1f B/clk/CU by s_memtime ticks (%lld ticks), %7.1f B/ns/CU by wall (%.3f ms)\n", name, threads, bytes / (double)h[0], h[0],
         bytes / (ms * 1e6), ms);
  hipFree(cyc); hipFree(sink);
}
int main() {
  for (int thr : {256, 512}) {
    run<0>("write b128", 16, thr);
    run<1>("write b64", 8, thr);
    run<2>("write b32", 4, thr);
    run<3>("read b128", 16, thr);
    run<4>("read b64", 8, thr);
  }
  return 0;
}
