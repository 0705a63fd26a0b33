// How fast can one CU pull GEMM-shaped operand tiles L2 -> LDS with global_load_lds, as a function of the number of
// K-tiles kept in flight?  Emulates gemm_bf16_kernel<64,2> traffic (A panel 128 x K, B panel 128 x K per block, 32 KiB per
// K-tile) without any MFMA/ds_read work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

template <int DEPTH>
__global__ __launch_bounds__(256, 2) void k(const uint16_t* A, const uint16_t* B, int K, int tiles_n, int* sink, int tiles_m, int mapping, int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  if (mapping) {
    const int bid = blockIdx.x, nwg = tiles_m * tiles_n;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int per_group = group_m * tiles_n;
    const int gidx = lid / per_group, first_m = gidx * group_m;
    const int gsz = min(tiles_m - first_m, group_m);
    const int in_g = lid - gidx * per_group;
    tm = first_m + in_g % gsz; tn = in_g / gsz;
  }
  const int nk = K / 64;
  auto stage = [&](int t, char* buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = wave * 4 + i, row = q * 8 + (lane >> 3), c = (lane & 7) ^ ((row >> 1) & 7);
      const uint16_t* ga = A + (size_t)(tm * 128 + row) * K + t * 64 + c * 8;
      const uint16_t* gb = B + (size_t)(tn * 128 + row) * K + t * 64 + c * 8;
      __builtin_amdgcn_global_load_lds(GLB_PTR(ga), (__attribute__((address_space(3))) void*)(buf + q * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(GLB_PTR(gb), (__attribute__((address_space(3))) void*)(buf + 16384 + q * 1024), 16, 0, 0);
    }
  };
  for (int t = 0; t < DEPTH - 1 && t < nk; ++t) stage(t, smem + (t % DEPTH) * 32768);
  for (int t = 0; t < nk; ++t) {
    if (t + DEPTH - 1 < nk) stage(t + DEPTH - 1, smem + ((t + DEPTH - 1) % DEPTH) * 32768);
    // wait until tile t has landed: leave (DEPTH-1) tiles (8 loads each per wave) in flight
    if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0 && smem[1] == 77 && smem[5000] == 1) *sink = 1;
}

template <int DEPTH>
void run(const uint16_t* A, const uint16_t* B, int M, int N, int K, int* sink, const char* tag, int mapping = 0, int group_m = 8) {
  const int tiles_m = M / 128, tiles_n = N / 128;
  const size_t lds = (size_t)DEPTH * 32768;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<DEPTH>, dim3(tiles_m * tiles_n), dim3(256), lds, 0, A, B, K, tiles_n, sink, tiles_m, mapping, group_m);
  CK(hipEventRecord(e0));
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<DEPTH>, dim3(tiles_m * tiles_n), dim3(256), lds, 0, A, B, K, tiles_n, sink, tiles_m, mapping, group_m);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  const double bytes = (double)tiles_m * tiles_n * (K / 64) * 32768.0;
  printf("map=%d g=%2d %-22s M=%5d N=%5d K=%5d depth=%d blocks=%5d  %8.1f us  %6.2f TB/s  %5.1f B/clk/CU(@2.4GHz)  equivalent GEMM %6.1f TF/s\n", mapping, group_m, tag, M, N, K, DEPTH,
         tiles_m * tiles_n, us, bytes / us / 1e6, bytes / us / 1e6 * 1e12 / 256 / 2.4e9, 2.0 * M * N * K / us / 1e6);
}

int main() {
  const size_t n = (size_t)250112 * 1024;
  uint16_t *A, *B; int* sink;
  CK(hipMalloc(&A, n * 2)); CK(hipMalloc(&B, n * 2)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(A, 0, n * 2)); CK(hipMemset(B, 0, n * 2));
  for (int g : {4, 8, 16, 32}) run<2>(A, B, 4096, 1024, 1024, sink, "so fwd (256 blocks)", 1, g);
  for (int g : {4, 8, 16, 32}) run<2>(A, B, 4096, 1024, 4096, sink, "fc2 fwd (256 blocks)", 1, g);
  for (int g : {4, 8, 16, 32}) run<2>(A, B, 4096, 4096, 4096, sink, "4096^3", 1, g);
  run<3>(A, B, 4096, 1024, 4096, sink, "fc2 fwd (256 blocks)", 1, 8);
  run<2>(A, B, 4096, 3072, 1024, sink, "qkv fwd (768 blocks)", 1, 8);
  run<2>(A, B, 4096, 250112, 1024, sink, "head fwd", 1, 8);
  run<2>(A, B, 4096, 250112, 1024, sink, "head fwd", 1, 32);
  run<1>(A, B, 4096, 1024, 1024, sink, "so fwd (256 blocks)");
  run<2>(A, B, 4096, 1024, 1024, sink, "so fwd (256 blocks)");
  run<3>(A, B, 4096, 1024, 1024, sink, "so fwd (256 blocks)");
  run<4>(A, B, 4096, 1024, 1024, sink, "so fwd (256 blocks)");
  run<2>(A, B, 4096, 1024, 4096, sink, "fc2 fwd (256 blocks)");
  run<4>(A, B, 4096, 1024, 4096, sink, "fc2 fwd (256 blocks)");
  run<2>(A, B, 4096, 4096, 4096, sink, "4096^3 (1024 blocks)");
  run<3>(A, B, 4096, 4096, 4096, sink, "4096^3 (1024 blocks)");
  run<4>(A, B, 4096, 4096, 4096, sink, "4096^3 (1024 blocks)");
  return 0;
}
