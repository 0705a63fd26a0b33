// probe_grid_barrier.hip — what a software grid barrier costs on this box next to a dependent kernel boundary, at the geometry of a
// decoder-step GEMM (256 workgroups of 1024 threads, 128 KiB of LDS each: one per CU).  Evidence for DESIGN.md §3 "Round 4" (the
// persistent decoder layer that was priced, not built).  Every spin is bounded: a barrier that does not complete within SPIN_CAP polls
// sets an error flag and the kernel returns (hang-proof; grid <= resident slots is checked by a census first).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe_grid_barrier tools/probe_grid_barrier.hip && /tmp/probe_grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define SPIN_CAP (1u << 22)

struct Sync {
  unsigned flat;        // monotonic arrival counter (flat barrier)
  unsigned pad0[31];
  unsigned grp[8][32];  // [g][0]: arrivals of logical group g = blockIdx % 8 (placement-independent: a group is a set of block ids)
  unsigned top;         // arrivals of the 8 group leaders
  unsigned pad1[31];
  unsigned gen[8][32];  // [g][0]: generation published by group g's leader
  unsigned err;         // set when a spin ran into its cap
  unsigned census;      // blocks that have started (residency check)
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// returns false (and sets err) when the cap is hit
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned want, unsigned* err) {
  unsigned n = 0;
  while (ld_relaxed(p) < want) {
    __builtin_amdgcn_s_sleep(2);
    if (++n > SPIN_CAP || ld_relaxed(err) != 0) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
  }
  return true;
}

// flat: one counter, everybody polls it
__device__ __forceinline__ bool barrier_flat(Sync* s, unsigned epoch, unsigned nblk) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&s->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ok = spin_until(&s->flat, epoch * nblk, &s->err);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// hierarchical: 8 logical groups (block id % 8 — on this stack that is also the XCD a block runs on, which makes it fast; nothing
// depends on it being so), group leader = the block with id < 8
__device__ __forceinline__ bool barrier_hier(Sync* s, unsigned epoch, unsigned nblk) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    const unsigned g = blockIdx.x & 7, members = (nblk - g + 7) / 8;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __hip_atomic_fetch_add(&s->grp[g][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x < 8) {
      ok = spin_until(&s->grp[g][0], epoch * members, &s->err);
      __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = ok && spin_until(&s->top, epoch * 8u, &s->err);
      __hip_atomic_store(&s->gen[g][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      ok = spin_until(&s->gen[g][0], epoch, &s->err);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

// MODE 0: flat, 1: hierarchical.  PAYLOAD 1: every thread publishes one 16-B word per phase (16 KiB per block) into the phase's half
// of a ping-pong buffer; after the barrier it reads the word its counterpart in block (b + 37) % nblk wrote and checks it (the other
// half is rewritten one phase later, i.e. behind the next barrier: one barrier per phase is enough).
template <int MODE, int PAYLOAD>
__global__ __launch_bounds__(1024) void barrier_kernel(Sync* s, uint4* buf, int nbar, unsigned nblk, unsigned* bad) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) lds[0] = 1;  // (the 128 KiB are only there to hold the CU)
  unsigned wrong = 0;
  const size_t half = (size_t)nblk * 1024;
  for (int i = 1; i <= nbar; ++i) {
    if (PAYLOAD) buf[(i & 1) * half + (size_t)blockIdx.x * 1024 + threadIdx.x] = make_uint4(i, blockIdx.x, threadIdx.x, 0x5eed);
    const bool ok = MODE == 0 ? barrier_flat(s, i, nblk) : barrier_hier(s, i, nblk);
    if (!ok) return;
    if (PAYLOAD) {
      const unsigned ob = (blockIdx.x + 37) % nblk;
      const uint4 v = buf[(i & 1) * half + (size_t)ob * 1024 + threadIdx.x];
      wrong += (v.x != (unsigned)i) | (v.y != ob) | (v.z != threadIdx.x);
    }
  }
  if (wrong) atomicAdd(bad, wrong);
}

__global__ __launch_bounds__(1024) void census_kernel(Sync* s, unsigned nblk) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) {
    lds[0] = 1;
    __hip_atomic_fetch_add(&s->census, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spin_until(&s->census, nblk, &s->err);  // completes only if all nblk blocks are resident at once
  }
}

template <int PAYLOAD>
__global__ __launch_bounds__(1024) void phase_kernel(uint4* buf, int i, unsigned nblk, unsigned* bad) {
  extern __shared__ char lds[];
  if (threadIdx.x == 0) lds[0] = 1;
  if (PAYLOAD) {
    // read what the PREVIOUS launch's counterpart wrote, then publish this phase's word (ping-pong halves of buf)
    const size_t half = (size_t)nblk * 1024;
    const unsigned ob = (blockIdx.x + 37) % nblk;
    if (i > 1) {
      const uint4 v = buf[((i - 1) & 1) * half + (size_t)ob * 1024 + threadIdx.x];
      if ((v.x != (unsigned)(i - 1)) | (v.y != ob) | (v.z != threadIdx.x)) atomicAdd(bad, 1u);
    }
    buf[(i & 1) * half + (size_t)blockIdx.x * 1024 + threadIdx.x] = make_uint4(i, blockIdx.x, threadIdx.x, 0x5eed);
  }
}

int main() {
  const unsigned nblk = 256;
  const int lds = 128 * 1024, NB = 400;
  Sync* s; uint4* buf; unsigned* bad;
  CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&buf, 2 * (size_t)nblk * 1024 * sizeof(uint4))); CK(hipMalloc(&bad, 4));
  CK(hipFuncSetAttribute((const void*)census_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void*)barrier_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void*)barrier_kernel<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void*)barrier_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void*)phase_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute((const void*)phase_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto reset = [&]() { CK(hipMemset(s, 0, sizeof(Sync))); CK(hipMemset(bad, 0, 4)); };
  auto err = [&]() { Sync h; CK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost)); return h.err; };
  // residency census: do 256 such blocks run at the same time?  (bounded: a short grid reports err instead of hanging)
  reset();
  hipLaunchKernelGGL(census_kernel, dim3(nblk), dim3(1024), lds, 0, s, nblk);
  CK(hipDeviceSynchronize());
  printf("census: %u blocks of 1024 threads + 128 KiB LDS resident together: %s\n", nblk, err() ? "NO (spin cap hit)" : "yes");
  if (err()) return 2;
  auto time_barrier = [&](auto kern, int nbar, const char* name) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      reset();
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(nblk), dim3(1024), lds, 0, s, buf, nbar, nblk, bad);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (err()) { printf("%s: spin cap hit (error flag set) — barrier did not complete\n", name); return -1.f; }
      best = ms < best ? ms : best;
    }
    unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    if (hb) printf("%s: %u WRONG words read after a barrier\n", name, hb);
    return best * 1e3f;
  };
  const float f0 = time_barrier(barrier_kernel<0, 0>, 0, "flat"), f1 = time_barrier(barrier_kernel<0, 0>, NB, "flat");
  const float h0 = time_barrier(barrier_kernel<1, 0>, 0, "hier"), h1 = time_barrier(barrier_kernel<1, 0>, NB, "hier");
  const float p1 = time_barrier(barrier_kernel<1, 1>, NB, "hier+16KiB");
  printf("flat counter barrier:        %.2f us per barrier (nothing published)\n", (f1 - f0) / NB);
  printf("8-group hierarchical barrier: %.2f us per barrier (nothing published)\n", (h1 - h0) / NB);
  printf("hierarchical, 16 KiB published per block and read by another block after it (checked): %.2f us per phase\n", (p1 - h0) / NB);
  // dependent kernel boundaries at the same geometry
  auto time_launches = [&](bool payload) {
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      reset();
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      for (int i = 1; i <= NB; ++i) {
        if (payload) hipLaunchKernelGGL(phase_kernel<1>, dim3(nblk), dim3(1024), lds, 0, buf, i, nblk, bad);
        else hipLaunchKernelGGL(phase_kernel<0>, dim3(nblk), dim3(1024), lds, 0, buf, i, nblk, bad);
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    unsigned hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    if (hb) printf("launch chain: %u WRONG words\n", hb);
    return best * 1e3f / NB;
  };
  printf("chain of %d dependent launches (same geometry), empty kernels:            %.2f us per launch\n", NB, time_launches(false));
  printf("chain of %d dependent launches, 16 KiB published / read per block per launch: %.2f us per launch\n", NB, time_launches(true));
  return 0;
}
