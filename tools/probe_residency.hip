// How many 256-thread blocks of a given dynamic-LDS size and register budget does the hardware keep RESIDENT on one CU at a time?
// Every block spins for a fixed time and records where it ran (XCC / SE / SH / CU) and when (wall_clock64); the host counts, per
// CU, the largest number of blocks whose intervals overlap.  The two-blocks-per-CU GEMMs (gemm_d2.hip: 72 KiB and 80 KiB of LDS,
// <= 256 registers) stand or fall with this number.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_residency.hip -o tools/probe_residency
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

template <int NREG>
__global__ __launch_bounds__(256, 2) void census(unsigned long long* out, long spin) {
  extern __shared__ char lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  float r[NREG];  // register ballast (kept live across the spin)
#pragma unroll
  for (int i = 0; i < NREG; ++i) r[i] = (float)(threadIdx.x + i);
  const unsigned long long w0 = wall_clock64();
  const long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < spin) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = r[i] * 1.0001f + 0.5f;
  }
  const unsigned long long w1 = wall_clock64();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NREG; ++i) s += r[i];
  if (s == 12345.678f) lds[threadIdx.x] = 1;  // (keeps r and the LDS allocation alive)
  if (threadIdx.x == 0) {
    out[3 * blockIdx.x] = ((unsigned long long)(xcc & 0xf) << 16) | (hw & 0xffff);
    out[3 * blockIdx.x + 1] = w0;
    out[3 * blockIdx.x + 2] = w1;
  }
}

template <int NREG>
static void run(int lds, int blocks) {
  unsigned long long* d;
  hipMalloc(&d, 3 * blocks * sizeof(unsigned long long));
  hipFuncSetAttribute(reinterpret_cast<const void*>(&census<NREG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(census<NREG>, dim3(blocks), dim3(256), lds, 0, d, 200000L);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed (lds %d)\n", lds); return; }
  std::vector<unsigned long long> h(3 * blocks);
  hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;  // per CU: (time, +1 / -1)
  for (int b = 0; b < blocks; ++b) {
    const unsigned long long id = h[3 * b], hw = id & 0xffff;
    const unsigned long long key = (id >> 16 << 16) | (hw & 0xff00);  // xcc | se, sh, cu (HW_ID bits 15:8)
    ev[key].push_back({h[3 * b + 1], +1});
    ev[key].push_back({h[3 * b + 2], -1});
  }
  int hist[9] = {0};
  for (auto& kv : ev) {
    auto& v = kv.second;
    std::sort(v.begin(), v.end());
    int cur = 0, mx = 0;
    for (auto& e : v) { cur += e.second; mx = std::max(mx, cur); }
    hist[std::min(mx, 8)]++;
  }
  printf("regs ~%3d  LDS %6d B  %4d blocks: %3zu CUs seen; CUs by max resident blocks:", NREG + 16, lds, blocks, ev.size());
  for (int i = 1; i <= 8; ++i) if (hist[i]) printf("  %d x%d", hist[i], i);
  printf("\n");
  hipFree(d);
}

int main() {
  for (int lds : {0, 65536, 73728, 81920}) run<64>(lds, 1024);
  for (int lds : {0, 65536, 73728, 81920}) run<170>(lds, 1024);
  return 0;
}
