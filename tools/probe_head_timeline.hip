// Where a block of the LM-head forward spends its time, and what its CU's other block does meanwhile.  Builds the two LDS-DMA GEMM
// kernels with -DMIC_TRACE_BLOCKS (four time stamps per block: entry, first operands landed, end of the K loop, stores acknowledged;
// wall_clock64 = one 100 MHz counter for the chip; plus the block's XCC / SE / CU) and runs the head shape [rows x 250112 x 1024] with
// softmax partials on gemm_d2 (256 x 128 tiles, two blocks per CU) and on gemm_w4 (256 x 256, one block per CU).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMIC_TRACE_BLOCKS -I multilingual-image-captioning_amd/csrc tools/probe_head_timeline.hip -o tools/probe_head_timeline
// usage: tools/probe_head_timeline [rows = 2432] [K = 1024]
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define mic_trace_buf mic_trace_buf_d2
#include "gemm_d2.hip"
#undef mic_trace_buf
#undef MIC_TRACE
#undef MIC_TRACE_ID
#define mic_trace_buf mic_trace_buf_w4
#include "gemm_w4.hip"
#undef mic_trace_buf

void mic_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int mic_cu_budget_now() { return 256; }

struct Blk { double t0, t1, t2, t3; unsigned long long cu; };

static void analyse(const char* name, const std::vector<unsigned long long>& h, int blocks) {
  std::vector<Blk> b(blocks);
  unsigned long long base = ~0ull;
  for (int i = 0; i < blocks; ++i) base = std::min(base, h[5 * i]);
  for (int i = 0; i < blocks; ++i) {
    const unsigned long long id = h[5 * i + 4];
    b[i] = {(h[5 * i] - base) * 0.01, (h[5 * i + 1] - base) * 0.01, (h[5 * i + 2] - base) * 0.01, (h[5 * i + 3] - base) * 0.01, (id >> 16 << 16) | (id & 0xff00)};
  }
  double end = 0;
  for (auto& x : b) end = std::max(end, x.t3);
  auto stat = [&](auto f, const char* what) {
    std::vector<double> v;
    for (auto& x : b) v.push_back(f(x));
    std::sort(v.begin(), v.end());
    double s = 0; for (double d : v) s += d;
    printf("  %-28s mean %7.2f  p10 %7.2f  median %7.2f  p90 %7.2f us\n", what, s / v.size(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
  };
  printf("%s: %d blocks, kernel span %.1f us\n", name, blocks, end);
  stat([](const Blk& x) { return x.t1 - x.t0; }, "entry -> first operands");
  stat([](const Blk& x) { return x.t2 - x.t1; }, "K loop");
  stat([](const Blk& x) { return x.t3 - x.t2; }, "epilogue (stores acked)");
  stat([](const Blk& x) { return x.t3 - x.t0; }, "whole block");
  // per CU: how much of the kernel's span has 0 / 1 / 2 blocks inside their K loop, and resident at all
  std::map<unsigned long long, std::vector<int>> cus;
  for (int i = 0; i < blocks; ++i) cus[b[i].cu].push_back(i);
  double k0 = 0, k1 = 0, k2 = 0, r0 = 0, r1 = 0, r2 = 0;
  for (auto& kv : cus) {
    std::vector<std::pair<double, int>> ev;  // (time, code): +-1 K loop, +-16 resident
    for (int i : kv.second) { ev.push_back({b[i].t1, 1}); ev.push_back({b[i].t2, -1}); ev.push_back({b[i].t0, 16}); ev.push_back({b[i].t3, -16}); }
    std::sort(ev.begin(), ev.end());
    int kin = 0, res = 0; double last = 0;
    for (auto& e : ev) {
      const double dt = e.first - last;
      (kin == 0 ? k0 : kin == 1 ? k1 : k2) += dt;
      (res == 0 ? r0 : res == 1 ? r1 : r2) += dt;
      last = e.first;
      if (e.second == 1 || e.second == -1) kin += e.second; else res += e.second / 16;
    }
    k0 += end - last; r0 += end - last;
  }
  const double tot = end * cus.size();
  printf("  %zu CUs; share of CU time with 0 / 1 / 2 blocks in their K loop: %.3f / %.3f / %.3f;  with 0 / 1 / 2 blocks resident: %.3f / %.3f / %.3f\n",
         cus.size(), k0 / tot, k1 / tot, k2 / tot, r0 / tot, r1 / tot, r2 / tot);
  // one CU's story
  auto& one = cus.begin()->second;
  std::vector<int> v(one.begin(), one.end());
  std::sort(v.begin(), v.end(), [&](int x, int y) { return b[x].t0 < b[y].t0; });
  printf("  one CU, its first blocks [entry, K loop from, to, stores acked] us:");
  for (size_t i = 0; i < v.size() && i < 10; ++i) printf("  [%.1f %.1f %.1f %.1f]", b[v[i]].t0, b[v[i]].t1, b[v[i]].t2, b[v[i]].t3);
  printf("\n");
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 2432, K = argc > 2 ? atoi(argv[2]) : 1024, N = 250112, V = 250054;
  uint16_t *A, *B, *C; float *bias, *stat;
  hipMalloc(&A, (size_t)4096 * K * 2); hipMalloc(&B, (size_t)N * K * 2); hipMalloc(&C, (size_t)4096 * N * 2);
  hipMalloc(&bias, (size_t)N * 4); hipMalloc(&stat, (size_t)4096 * (N / 64) * 8);
  hipMemset(A, 0x3c, (size_t)4096 * K * 2); hipMemset(B, 0x3c, (size_t)N * K * 2); hipMemset(bias, 0, (size_t)N * 4);
  for (int which = 0; which < 4; ++which) {
    const bool d2 = which < 2, stats = (which & 1) == 0;
    LaunchTable tab{};
    tab.count = 1;
    Problem& p = tab.p[0];
    p.A = A; p.B = B; p.lda = K; p.ldb = K; p.M = M; p.N = N; p.K = K; p.nsplit = 1; p.k_valid = 0x7fffffff;
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + (d2 ? 127 : 255)) / (d2 ? 128 : 256);
    p.epi.C = C; p.epi.ldc = N; p.epi.bias = bias; p.epi.alpha = 1.0f; p.epi.N = N;
    if (stats) { p.epi.rowstat = stat; p.epi.stat_ld = N / 64; p.epi.stat_nvalid = V; }
    tab.total_blocks = p.tiles_m * p.tiles_n;
    unsigned long long* d;
    hipMalloc(&d, (size_t)5 * tab.total_blocks * 8);
    if (d2) hipMemcpyToSymbol(HIP_SYMBOL(mic_trace_buf_d2), &d, sizeof(d)); else hipMemcpyToSymbol(HIP_SYMBOL(mic_trace_buf_w4), &d, sizeof(d));
    for (int rep = 0; rep < 2; ++rep) {  // (the second launch is the one read)
      if (d2) launch_gemm_d2(tab, 0); else launch_gemm_w4(tab, 0);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
    }
    std::vector<unsigned long long> h((size_t)5 * tab.total_blocks);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    char name[128];
    snprintf(name, sizeof name, "%s, %d x %d x %d, bf16 C + bias%s", d2 ? "gemm_d2 (256 x 128, two per CU)" : "gemm_w4 (256 x 256)", M, N, K, stats ? " + softmax partials" : "");
    analyse(name, h, tab.total_blocks);
    hipFree(d);
  }
  return 0;
}
