// Streams with a CU mask (hipExtStreamCreateWithCUMask) on this stack: where do the blocks of a masked stream land (XCC / SE / CU
// ids from the hardware registers), what streaming bandwidth do n CUs reach on an AdamW-shaped pass (4 fp32 streams in, 3 out), and
// does a full-chip spin kernel on an unmasked stream run beside it?
// build: hipcc --offload-arch=gfx950 -O2 tools/probe_cu_mask.hip -o tools/probe_cu_mask
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <set>
#include <vector>

__global__ void where(unsigned* out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < 20000) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

__global__ __launch_bounds__(256) void stream7(size_t n4, const float4* __restrict__ p, const float4* __restrict__ m, const float4* __restrict__ v,
                                                const float4* __restrict__ g, float4* po, float4* mo, float4* vo) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 a = p[i], b = m[i], c = v[i], d = g[i];
    b.x = 0.9f * b.x + 0.1f * d.x; b.y = 0.9f * b.y + 0.1f * d.y; b.z = 0.9f * b.z + 0.1f * d.z; b.w = 0.9f * b.w + 0.1f * d.w;
    c.x = 0.99f * c.x + 0.01f * d.x * d.x; c.y = 0.99f * c.y + 0.01f * d.y * d.y; c.z = 0.99f * c.z + 0.01f * d.z * d.z; c.w = 0.99f * c.w + 0.01f * d.w * d.w;
    a.x -= 1e-3f * b.x * __frsqrt_rn(c.x + 1e-8f); a.y -= 1e-3f * b.y * __frsqrt_rn(c.y + 1e-8f); a.z -= 1e-3f * b.z * __frsqrt_rn(c.z + 1e-8f); a.w -= 1e-3f * b.w * __frsqrt_rn(c.w + 1e-8f);
    po[i] = a; mo[i] = b; vo[i] = c;
  }
}

__global__ void spin(long cycles) {
  extern __shared__ char lds[];
  const long t0 = __builtin_readcyclecounter();
  while (__builtin_readcyclecounter() - t0 < cycles) {}
}

static hipStream_t masked(const std::vector<int>& cus) {
  uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int c : cus) mask[c >> 5] |= 1u << (c & 31);
  hipStream_t s = nullptr;
  hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
  if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(e)); return nullptr; }
  return s;
}

int main() {
  hipDeviceProp_t pr;
  hipGetDeviceProperties(&pr, 0);
  printf("multiProcessorCount %d\n", pr.multiProcessorCount);
  unsigned* d;
  hipMalloc(&d, 2 * 4096 * sizeof(unsigned));
  std::vector<unsigned> h(2 * 4096);
  auto placement = [&](const char* name, hipStream_t s) {
    hipLaunchKernelGGL(where, dim3(4096), dim3(64), 0, s, d);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::set<unsigned> cus;
    int per_xcc[8] = {0};
    for (int i = 0; i < 4096; ++i) {
      const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
      const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;  // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
      if (cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu).second) per_xcc[xcc & 7]++;
    }
    printf("%-40s distinct (xcc, se, sh, cu): %3zu   per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
  };
  hipStream_t s0;
  hipStreamCreate(&s0);
  placement("unmasked stream", s0);
  std::vector<int> first32, first64, every8th, low4;
  for (int i = 0; i < 32; ++i) first32.push_back(i);
  for (int i = 0; i < 64; ++i) first64.push_back(i);
  for (int i = 0; i < 256; i += 8) every8th.push_back(i);
  for (int i = 0; i < 256; ++i) if ((i & 31) < 4) low4.push_back(i);
  hipStream_t m32 = masked(first32), m64 = masked(first64), m8 = masked(every8th), ml4 = masked(low4);
  if (!m32) return 1;
  placement("mask: CUs 0..31", m32);
  placement("mask: CUs 0..63", m64);
  placement("mask: every 8th CU (32 bits)", m8);
  placement("mask: CUs with (i & 31) < 4 (32 bits)", ml4);

  // AdamW-shaped streaming pass
  const size_t n = (size_t)64 << 20;  // 64 M floats per array = 256 MB, 7 arrays
  float *p, *m, *v, *g;
  hipMalloc(&p, n * 4); hipMalloc(&m, n * 4); hipMalloc(&v, n * 4); hipMalloc(&g, n * 4);
  hipMemset(p, 0, n * 4); hipMemset(m, 0, n * 4); hipMemset(v, 0, n * 4); hipMemset(g, 0, n * 4);
  auto bw = [&](const char* name, hipStream_t s, int blocks) {
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
      hipDeviceSynchronize();
      auto t0 = std::chrono::high_resolution_clock::now();
      hipLaunchKernelGGL(stream7, dim3(blocks), dim3(256), 0, s, n / 4, (const float4*)p, (const float4*)m, (const float4*)v, (const float4*)g, (float4*)p, (float4*)m, (float4*)v);
      hipStreamSynchronize(s);
      const double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
      if (us < best) best = us;
    }
    printf("%-40s %5d blocks: %8.1f us  %6.2f TB/s\n", name, blocks, best, 7.0 * n * 4 / best * 1e-6);
    return best;
  };
  bw("unmasked", s0, 8192);
  for (int blocks : {256, 1024, 8192}) {
    bw("mask 0..31", m32, blocks);
    bw("mask 0..63", m64, blocks);
    bw("mask every 8th", m8, blocks);
    bw("mask (i&31)<4", ml4, blocks);
  }
  // a full-chip spin kernel (256 blocks x 1024 threads, 128 KB LDS) beside the masked streaming pass
  hipFuncSetAttribute(reinterpret_cast<const void*>(&spin), hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  for (hipStream_t ms : {m32, ml4}) {
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    hipLaunchKernelGGL(stream7, dim3(1024), dim3(256), 0, ms, n / 4, (const float4*)p, (const float4*)m, (const float4*)v, (const float4*)g, (float4*)p, (float4*)m, (float4*)v);
    for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(spin, dim3(224), dim3(1024), 131072, s0, 400000L);
    hipStreamSynchronize(s0);
    const double t_spin = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    hipDeviceSynchronize();
    const double t_all = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
    printf("masked streaming pass + 8 spin launches of 224 blocks x 1024 threads x 128 KB on the unmasked stream: spins done %8.1f us, all done %8.1f us\n", t_spin, t_all);
  }
  hipDeviceSynchronize();
  auto t0 = std::chrono::high_resolution_clock::now();
  for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(spin, dim3(224), dim3(1024), 131072, s0, 400000L);
  hipDeviceSynchronize();
  printf("the 8 spin launches alone: %8.1f us\n", std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count());
  return 0;
}
